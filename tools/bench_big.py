#!/usr/bin/env python3
"""S-big: the HBM-bound form of the BPR hot path (BASELINE.json configs[4] on one GPU).

Synthetic U = 10 M users x I = 1 M items, d = 128 fp32 (tables + Adam slots = 16.9 GB, far
beyond the 256 MiB Infinity Cache), batch 6144, lazy Adam (only touched rows move; a TF1-exact
dense sweep would stream all 16.9 GB per step).  Users uniform, positives Zipf(1.0), negatives
uniform.  Measures the two kernels of the lazy step (triple pass, multi-occurrence rows) with HIP events and prices them against
SURVEY.md 8(d)'s algorithmic bytes: (24d+12) per triple + 28d per unique touched row.

    python tools/bench_big.py [--users 10000000] [--items 1000000] [--dim 128] [--steps 300]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(users=10_000_000, items=1_000_000, dim=128, steps=300, warmup=30, batch=6144, sess=None, seed=0, zipf=True):
    import torch
    from cleverrec_amd import _lib, runtime
    sess = sess or runtime.Session(0)
    dev = sess.device
    g = torch.Generator(device=dev).manual_seed(seed)
    st = runtime.BprState.__new__(runtime.BprState)
    st.sess = sess
    f32 = torch.float32
    st.P = torch.empty((users, dim), dtype=f32, device=dev).normal_(0, 0.01, generator=g)
    st.Q = torch.empty((items, dim), dtype=f32, device=dev).normal_(0, 0.01, generator=g)
    st.mP, st.vP = torch.zeros_like(st.P), torch.zeros_like(st.P)
    st.mQ, st.vQ = torch.zeros_like(st.Q), torch.zeros_like(st.Q)
    st.tables = _lib.BprTables(*(t.data_ptr() for t in (st.P, st.Q, st.mP, st.vP, st.mQ, st.vQ)), users, items, dim)
    st.adam = _lib.Adam(1e-3, 0.9, 0.999, 1e-8, 0.01, _lib.ADAM_LAZY, 0.9, 0.999, 0)

    n = (steps + warmup) * batch
    u = torch.randint(0, users, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    w = 1.0 / torch.arange(1, items + 1, dtype=torch.float64, device=dev)
    cdf = torch.cumsum(w / w.sum(), 0)
    i = torch.searchsorted(cdf, torch.rand(n, generator=g, device=dev, dtype=torch.float64)).clamp_(max=items - 1).to(torch.int32)
    if not zipf:  # experiment: uniform positives (hardly any row occurs twice in a batch: the triple pass's main line alone)
        i = torch.randint(0, items, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    j = torch.randint(0, items, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    del cdf, w

    nw = warmup * batch
    st.train_epoch(u[:nw], i[:nw], j[:nw], batch)
    st.train_epoch(u[nw:], i[nw:], j[nw:], batch)  # the timed call's shape: every workspace (a second index buffer, the side stream) reaches its final size
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = st.train_epoch(u[nw:], i[nw:], j[nw:], batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0

    sess.ctx.profile(True)
    st.train_epoch(u[nw:], i[nw:], j[nw:], batch)
    torch.cuda.synchronize()
    prof = sess.ctx.profile_read()
    batches_per_launch = sess.ctx.profile_units("bpr_triples_lazy") / max(1, prof["bpr_triples_lazy"][1])  # (option lazy_window: W batches per launch)
    sess.ctx.profile(False)

    # row multiplicities per batch (sampled): unique touched rows for the 28d Adam term, and which kernel owns a row --
    # rows that occur once are finished in the triple pass's main line, the others by their last contributor
    uq_u = uq_q = single_u = single_q = multi_slots_q = multi_slots_u = 0
    sample = range(0, steps, max(1, steps // 16))
    for b in sample:
        sl = slice(nw + b * batch, nw + (b + 1) * batch)
        cu = torch.unique(u[sl], return_counts=True)[1]
        cq = torch.unique(torch.cat([i[sl], j[sl]]), return_counts=True)[1]
        uq_u += int(cu.numel())
        uq_q += int(cq.numel())
        single_u += int((cu == 1).sum())
        single_q += int((cq == 1).sum())
        multi_slots_u += int(cu[cu > 1].sum())
        multi_slots_q += int(cq[cq > 1].sum())
    uq_u, uq_q, single_u, single_q = (x / len(sample) for x in (uq_u, uq_q, single_u, single_q))
    multi_slots_u, multi_slots_q = multi_slots_u / len(sample), multi_slots_q / len(sample)
    d = dim
    # (24d + 12) B per triple (3 ids, 3 row reads, 3 gradient rows) + 28d B per unique touched row (Adam).
    # The step is ONE launch (bpr_triples_lazy): it owns all of it.
    alg_step = batch * (24 * d + 12) + (uq_u + uq_q) * 28 * d
    us = prof["bpr_triples_lazy"][0] / prof["bpr_triples_lazy"][1] * 1e3 / batches_per_launch  # per batch
    out = {
        "workload": "S-big: %d users x %d items, d=%d fp32, batch %d, lazy Adam, Zipf(1.0) positives" % (users, items, dim, batch),
        "state_GB": (users + items) * dim * 4 * 3 / 1e9,
        "triples_per_s": steps * batch / dt, "us_per_step": dt / steps * 1e6,
        "unique_user_rows_per_batch": uq_u, "unique_item_rows_per_batch": uq_q,
        "single_occurrence_rows_per_batch": {"users": single_u, "items": single_q},
        "slots_of_multi_occurrence_rows_per_batch": {"users": multi_slots_u, "items": multi_slots_q},
        "bpr_triples_lazy": {"avg_us": us, "batches_per_launch": batches_per_launch, "algorithmic_bytes": alg_step, "GBs": alg_step / us / 1e3, "frac_of_8TBs": alg_step / us / 1e3 / 8000},
        "step_algorithmic_bytes": alg_step,
        "step_algorithmic_GBs": alg_step / (dt / steps) / 1e9,
        "bpr_index_batches_us_per_step": prof.get("bpr_index_batches", (0.0, 0))[0] * 1e3 / steps,  # (0: batches beyond 8192 are indexed by index_big.hip, outside these brackets)
        "loss_first_last": [float(loss[0]), float(loss[-1])],
    }
    del st
    torch.cuda.empty_cache()
    return out


def run_sharded(users=10_000_000, items=1_000_000, dim=128, steps=200, warmup=30, batch=6144, sess=None, seed=0, rank=0,
                world=1, barrier=None, max_reduce=None):
    """The same workload row-sharded over `world` ranks (crec_bpr_train_epoch_sharded): users range-sharded
    (users / world rows of P per rank), items dealt round-robin (ceil(items / world) rows of Q per rank), `batch`
    triples per rank per step (weak scaling in triples, the tables' total size fixed).  Collective: every rank calls it.
    barrier(): device sync + process barrier; max_reduce(x): max over ranks of a host float.  world == 1 measures
    what the exchange protocol itself costs against the direct lazy step (plan, compact table, serve / apply)."""
    import torch
    from cleverrec_amd import parallel, runtime
    sess = sess or runtime.Session(0)
    dev = sess.device
    barrier = barrier or torch.cuda.synchronize
    max_reduce = max_reduce or (lambda x: x)
    transport = "the ctx's communicator" if world > 1 else "none (device copies)"
    if world == 1 and os.environ.get("CREC_TRANSPORT") == "peer":  # a world of one over the peer windows: what the fused exchange kernels cost by themselves
        if not getattr(sess, "peer_window_bytes", 0):  # (bench.py's rehearsal of the N > 1 path has connected this ctx already)
            parallel.init_peer(sess, 0, 1, window_bytes=max(64 << 20, -(-(2 * (2 * batch * dim * 4 + 4096)) // (1 << 20)) * (1 << 20)))  # two halves, each a batch's compact table
        transport = "peer windows, a world of one" + (" (five launches per step)" if os.environ.get("CREC_SHARD_UNFUSED") == "1" else " (exchanges fused with serve / apply)")
    g = torch.Generator(device=dev).manual_seed(seed + 1000 * rank)
    lo, hi = parallel.user_range(users, rank, world)
    S = parallel.shard_rows(items, world)
    f32 = torch.float32
    st = parallel.ShardedBprState.__new__(parallel.ShardedBprState)
    st.sess = sess
    st.P = torch.empty((hi - lo, dim), dtype=f32, device=dev).normal_(0, 0.01, generator=g)
    st.Q = torch.empty((S, dim), dtype=f32, device=dev).normal_(0, 0.01, generator=g)
    st.mP, st.vP = torch.zeros_like(st.P), torch.zeros_like(st.P)
    st.mQ, st.vQ = torch.zeros_like(st.Q), torch.zeros_like(st.Q)
    from cleverrec_amd import _lib
    st.tables = _lib.BprTables(*(t.data_ptr() for t in (st.P, st.Q, st.mP, st.vP, st.mQ, st.vQ)), hi - lo, S, dim)
    st.adam = _lib.Adam(1e-3, 0.9, 0.999, 1e-8, 0.01, _lib.ADAM_LAZY, 0.9, 0.999, 0)
    st.item_nums, st.rank, st.world, st.stats = items, rank, world, _lib.ShardStats()

    n = steps * batch
    u = torch.randint(0, hi - lo, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    w = 1.0 / torch.arange(1, items + 1, dtype=torch.float64, device=dev)
    cdf = torch.cumsum(w / w.sum(), 0)
    i = torch.searchsorted(cdf, torch.rand(n, generator=g, device=dev, dtype=torch.float64)).clamp_(max=items - 1).to(torch.int32)
    j = torch.randint(0, items, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    del cdf, w
    barrier()  # (every rank has filled its tables and drawn its triples: nobody's exchange kernel waits through a peer's set-up)
    st.train_epoch(u, i, j, batch)  # same call shape as the timed one: every workspace reaches its final size
    barrier()
    st.stats = _lib.ShardStats()
    t0 = time.perf_counter()
    loss = st.train_epoch(u, i, j, batch)
    barrier()
    dt = max_reduce(time.perf_counter() - t0)
    stats = st.stats_dict()
    sess.ctx.profile(True)
    st.train_epoch(u, i, j, batch)
    barrier()
    prof = sess.ctx.profile_read()
    sess.ctx.profile(False)
    kern = {k: round(ms / cnt * 1e3, 2) for k, (ms, cnt) in prof.items() if cnt}
    per_step = {k: stats[k] / steps for k in ("bytes_ids_sent", "bytes_rows_sent", "bytes_grads_sent", "rows_requested", "rows_served")}
    out = {
        "workload": "S-big row-sharded: %d users x %d items over %d rank(s), d=%d fp32, batch %d per rank per step, lazy Adam, "
                    "Zipf(1.0) positives" % (users, items, world, dim, batch),
        "world": world, "transport": transport, "triples_per_s": steps * batch * world / dt, "us_per_step": dt / steps * 1e6, "steps": steps,
        "exchange_per_rank_per_step": per_step, "max_rows_served_per_batch": stats["max_rows_served_per_batch"],
        "kernel_avg_us": kern, "loss_first_last": [float(loss[0]), float(loss[-1])],
        "exchange": "per chunk: request lists (all-to-all-v); per step: rows forward, gradient rows back (2 all-to-all-v)",
    }
    del st
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=10_000_000)
    ap.add_argument("--items", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=6144)
    ap.add_argument("--uniform", action="store_true", help="uniform positives instead of Zipf(1.0): hardly any multi-occurrence rows")
    ap.add_argument("--sharded", action="store_true", help="the row-sharded epoch on a world of one (protocol cost)")
    a = ap.parse_args()
    if a.sharded:
        print(json.dumps(run_sharded(a.users, a.items, a.dim, a.steps, batch=a.batch)))
    else:
        print(json.dumps(run(a.users, a.items, a.dim, a.steps, batch=a.batch, zipf=not a.uniform)))
