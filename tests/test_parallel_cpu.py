"""The multi-GPU paths' host logic and exchange protocol on CPU: gloo processes, world sizes 2 and 4.

What can be checked without GPUs: the user-range sharding of the `data` protocol, the byte/int
agreement helpers the RCCL bootstrap uses, and -- with the CPU oracle standing in for the kernels --
that the exchange scheme itself (per-rank dQ partials -> all-reduce -> identical dense Adam on every
replica, crec_bpr_train_epoch_dp) reproduces the single-process step on the concatenated batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from cleverrec_amd import parallel
from cleverrec_amd.data import RankingData


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _toy(seed=0, U=41, I=30):
    g = np.random.Generator(np.random.PCG64(seed))
    ui_train = {u: [int(x) for x in g.choice(I, size=int(g.integers(2, 9)), replace=False)] for u in range(U) if u % 9 != 8}
    ui_test = {u: [int(x) for x in g.choice(I, size=2, replace=False)] for u in range(0, U, 3)}
    return RankingData(U, I, ui_train, ui_test)


def test_user_range_and_shards_partition_the_data():
    data = _toy()
    for world in (1, 2, 3, 8):
        seen_users, n_train, n_test = [], 0, 0
        for r in range(world):
            lo, hi = parallel.user_range(data.user_nums, r, world)
            sh = parallel.shard_users(data, r, world)
            assert sh.user_nums == hi - lo and sh.item_nums == data.item_nums
            seen_users += list(range(lo, hi))
            n_train += sh.n_train
            n_test += len(sh.test_csr[1])
            for k, (uu, items) in enumerate(sh.ui_train.items()):
                assert data.ui_train[uu + lo] == items
            for uu, items in sh.ui_test.items():
                assert data.ui_test[uu + lo] == items
        assert seen_users == list(range(data.user_nums))
        assert n_train == data.n_train and n_test == len(data.test_csr[1])


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # bootstrap helpers
        assert parallel.broadcast_bytes(b"\x01\x02id-from-rank0" if rank == 0 else b"", 0) == b"\x01\x02id-from-rank0"
        assert parallel.agree_min(10 + rank) == 10
        # sharded evaluation: metric sums of the ranks' own users -> means over all users; item shards -> the full table
        sums = torch.tensor([[1.0 + rank, 2.0], [0.5, 0.25 * rank], [3.0, 1.0]], dtype=torch.float64)
        means, total = parallel.reduce_metric_sums(sums, 3 + rank)
        assert total == 7 and torch.allclose(means, torch.tensor([[3.0, 4.0], [1.0, 0.25], [6.0, 2.0]], dtype=torch.float64) / 7)
        Qfull = torch.arange(5 * 3, dtype=torch.float32).reshape(5, 3)
        got = parallel.gather_item_shards(torch.from_numpy(parallel.shard_items(Qfull.numpy(), rank, world)), 5)
        assert torch.equal(got, Qfull)

        # the exchange scheme, oracle as the per-rank compute
        data = _toy()
        d, B, steps = 8, 64, 3
        g = np.random.Generator(np.random.PCG64(5))
        P0 = (g.standard_normal((data.user_nums, d)) * 0.1).astype(np.float32)
        Q0 = (g.standard_normal((data.item_nums, d)) * 0.1).astype(np.float32)
        lo, hi = parallel.user_range(data.user_nums, rank, world)
        shard = parallel.shard_users(data, rank, world)
        m = oracle.BprModel(P0[lo:hi], Q0, lr=1e-2, reg=0.01)
        rng = oracle.Rng(100 + rank)  # every rank samples its own users
        _, u, i, j = oracle.sample_pairwise(rng, shard.ui_train, shard.item_nums, 2, B)
        losses = []
        for s in range(steps):
            sl = slice(s * B, (s + 1) * B)
            lr_t = float(m.sched.next(1)[0])
            dQ, loss = oracle.bpr_dp_local(m, u[sl], i[sl], j[sl], lr_t)
            t = torch.from_numpy(dQ)
            dist.all_reduce(t)
            oracle.adam_dense(m, t.numpy(), lr_t)
            lt = torch.tensor([loss], dtype=torch.float64)
            dist.all_reduce(lt)
            losses.append(float(lt))
        # replicas of Q are bit-identical
        allq = [torch.zeros_like(torch.from_numpy(m.Q)) for _ in range(world)]
        dist.all_gather(allq, torch.from_numpy(m.Q))
        assert all(torch.equal(allq[0], q) for q in allq)
        gathered = [None] * world
        dist.all_gather_object(gathered, (lo, m.P, u[:steps * B] + lo, i[:steps * B], j[:steps * B]))
        if rank == 0:
            # single-process run on the concatenated batches
            ref = oracle.BprModel(P0, Q0, lr=1e-2, reg=0.01)
            ref_losses = []
            for s in range(steps):
                sl = slice(s * B, (s + 1) * B)
                uu = np.concatenate([x[2][sl] for x in gathered])
                ii = np.concatenate([x[3][sl] for x in gathered])
                jj = np.concatenate([x[4][sl] for x in gathered])
                ref_losses.append(ref.step(uu, ii, jj))
            np.testing.assert_allclose(losses, ref_losses, rtol=1e-6)
            np.testing.assert_allclose(m.Q, ref.Q, rtol=0, atol=2e-6)
            for lo_r, P_r, *_ in gathered:
                np.testing.assert_allclose(P_r, ref.P[lo_r:lo_r + len(P_r)], rtol=0, atol=2e-6)
            out.put("ok")
    finally:
        dist.destroy_process_group()


def test_exchange_scheme_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) == "ok"


class _LibPlanner:
    """the product's host planner (libcrec.so: csrc/shard_host.cpp, same arithmetic and orders as the device kernels)"""
    plan = staticmethod(parallel.plan_batch_host)
    owner_index = staticmethod(parallel.owner_index_host)


def _shard_worker(rank, world, port, out, use_lib):
    """The row-sharded exchange (include/crec.h "row-sharded training") as real multi-process all-to-all-v over gloo:
    every rank holds ONLY its user rows and its item shard (oracle/shard_protocol.py); the result must equal, bit for
    bit, the single-process restatement of the sharded arithmetic (oracle.sharded_steps: per-rank sums, then rank
    order), i.e. owner mapping, compact ids, request lists and segment offsets all line up."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import shard_protocol as SP
        U, I, d, B, steps = 97, 53, 8, 64, 3  # 53 items: no world size divides it, the last shard rows are padding
        g = np.random.Generator(np.random.PCG64(7))
        P0 = (g.standard_normal((U, d)) * 0.1).astype(np.float32)
        Q0 = (g.standard_normal((I, d)) * 0.1).astype(np.float32)
        ranges = [parallel.user_range(U, r, world) for r in range(world)]
        streams = []
        for lo, hi in ranges:  # every rank can rebuild every stream (same generator): rank 0 needs them all
            n = steps * B - int(g.integers(0, B // 2))
            streams.append((g.integers(0, hi - lo, n).astype(np.int32), np.minimum(g.zipf(1.5, n) - 1, I - 1).astype(np.int32),
                            g.integers(0, I, n).astype(np.int32)))
        lo, hi = ranges[rank]
        me = SP.ShardedRank(P0[lo:hi], parallel.shard_items(Q0, rank, world), I, lr=1e-2, reg=0.01,
                            planner=_LibPlanner if use_lib else None)
        u, i, j = streams[rank]
        losses = [me.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], j[b * B:(b + 1) * B]) for b in range(steps)]
        gathered = [None] * world
        dist.all_gather_object(gathered, (me.user.P, me.item.Q))
        if rank == 0:
            P = np.concatenate([x[0] for x in gathered])
            Q = parallel.unshard_items([x[1] for x in gathered], I)
            oP, oQ, oloss, _ = oracle.sharded_steps(P0, Q0, ranges, streams, B, lr=1e-2, reg=0.01)
            assert np.array_equal(P, oP) and np.array_equal(Q, oQ)
            np.testing.assert_allclose(losses, oloss, rtol=1e-12)
            # and the same global batches on ONE rank (the plain lazy step on the rank-major concatenation): fp32 add order only
            ref = oracle.BprModel(P0, Q0, lr=1e-2, reg=0.01, adam_mode=1)
            for b in range(steps):
                parts = [(s[0][b * B:(b + 1) * B] + r[0], s[1][b * B:(b + 1) * B], s[2][b * B:(b + 1) * B]) for s, r in zip(streams, ranges)]
                ref.step(*(np.concatenate([p[k] for p in parts]) for k in range(3)))
            np.testing.assert_allclose(P, ref.P, rtol=0, atol=5e-6)
            np.testing.assert_allclose(Q, ref.Q, rtol=0, atol=5e-6)
            out.put("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("use_lib", [True, False], ids=["library-planner", "numpy-planner"])
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_exchange_protocol_gloo(world, use_lib):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, out, use_lib)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert out.get(timeout=5) == "ok"
