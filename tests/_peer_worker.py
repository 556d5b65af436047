"""One rank of the multi-PROCESS peer-transport tests (tests/test_gpu_peer.py starts WORLD_SIZE of these on cuda:0; on a node
the same code runs one per GPU).  gloo carries the IPC handles and the results; every payload crosses through the windows."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle  # noqa: E402
from cleverrec_amd import parallel, runtime  # noqa: E402


def exchanges(sess, rank, world, rounds=7):
    """random all-to-all-v rounds (ragged, empty and unaligned segments; back to back, so both window halves and the flag
    parities are reused) and all-reduces -> the received words, checked by rank 0 against what everybody says it sent"""
    import ctypes as C
    out = []
    for r in range(rounds):
        g = np.random.Generator(np.random.PCG64(1000 + r))  # the same matrix of counts on every rank
        cnt = g.integers(0, 5000, (world, world))
        cnt[g.random((world, world)) < 0.2] = 0
        if r % 2:
            cnt *= 4  # 16-byte multiples: the vector path
        mine = [np.random.Generator(np.random.PCG64(77 * r + 13 * rank + q)).integers(0, 2 ** 31, int(cnt[rank, q])).astype(np.int32) for q in range(world)]
        send = np.concatenate(mine) if sum(len(m) for m in mine) else np.zeros(1, np.int32)
        scnt = cnt[rank].astype(np.int64)
        sdsp = np.concatenate([[0], np.cumsum(scnt)[:-1]]).astype(np.int64)
        rcnt = cnt[:, rank].astype(np.int64)
        rdsp = np.concatenate([[0], np.cumsum(rcnt)[:-1]]).astype(np.int64)
        d_send, d_recv = sess.to_dev(send), sess.zeros(int(rcnt.sum()) + 1, torch.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        sess.ctx.check(sess.lib.crec_debug_alltoallv(sess.ctx.h, sess.ptr(d_send), p(scnt), p(sdsp), sess.ptr(d_recv), p(rcnt), p(rdsp), sess.stream_ptr()))
        got = d_recv.cpu().numpy()[:int(rcnt.sum())]
        want = np.concatenate([np.random.Generator(np.random.PCG64(77 * r + 13 * q + rank)).integers(0, 2 ** 31, int(cnt[q, rank])).astype(np.int32)
                               for q in range(world)]) if rcnt.sum() else np.zeros(0, np.int32)
        out.append(bool(np.array_equal(got, want)))
        n = [1, 37, 4096, 237184][r % 4]
        x = np.random.Generator(np.random.PCG64(5 * r + rank)).standard_normal(n).astype(np.float32)
        d_x = sess.to_dev(x)
        sess.ctx.check(sess.lib.crec_debug_allreduce(sess.ctx.h, sess.ptr(d_x), n, sess.stream_ptr()))
        acc = np.zeros(n, np.float32)
        for q in range(world):  # rank order
            acc = acc + np.random.Generator(np.random.PCG64(5 * r + q)).standard_normal(n).astype(np.float32)
        out.append(bool(np.array_equal(d_x.cpu().numpy(), acc)))
    sess.ctx.check(sess.lib.crec_comm_status(sess.ctx.h, sess.stream_ptr()))
    return out


def sharded(sess, rank, world):
    """the row-sharded epoch, every rank a process: tables and losses against the oracle's restatement (as tests/test_gpu_sharded.py)"""
    U, I, B, d = 400, 301, 256, 32
    g = np.random.Generator(np.random.PCG64(3))
    P0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32)
    Q0 = (g.standard_normal((I, d)) * 0.01).astype(np.float32)
    ranges = [parallel.user_range(U, r, world) for r in range(world)]
    streams = []
    for r, (lo, hi) in enumerate(ranges):
        n = 2 * B + [40, 7, 1, 256][r % 4]
        streams.append((g.integers(0, hi - lo, n).astype(np.int32), np.minimum(g.zipf(1.3, n) - 1, I - 1).astype(np.int32), g.integers(0, I, n).astype(np.int32)))
    lo, hi = ranges[rank]
    st = parallel.ShardedBprState(sess, P0[lo:hi], parallel.shard_items(Q0, rank, world), I, 1e-3, 0.01, rank, world)
    u, i, j = (sess.to_dev(x) for x in streams[rank])
    loss = st.train_epoch(u, i, j, B).cpu().numpy()
    torch.cuda.synchronize()
    gathered = [None] * world
    dist.all_gather_object(gathered, (st.P.cpu().numpy(), st.Q.cpu().numpy(), loss))
    if rank:
        return None
    P = np.concatenate([x[0] for x in gathered])
    Q = parallel.unshard_items([x[1] for x in gathered], I)
    oP, oQ, oloss, _ = oracle.sharded_steps(P0, Q0, ranges, streams, B)
    return {"loss_ok": bool(np.allclose(loss, oloss, rtol=1e-5, atol=0)), "P_err": float(np.abs(P - oP).max()), "Q_err": float(np.abs(Q - oQ).max()),
            "same_loss_everywhere": all(bool(np.array_equal(x[2], loss)) for x in gathered)}


def data_parallel(sess, rank, world, I=140):
    """the replicated-Q step, every rank a process: replicas bit-identical, tables against the oracle's restatement.  Over the peer
    transport the all-reduce of dQ and the dense Adam are ONE launch (option dp_unfused: two); I = 300: its two-shot form at a world of 3"""
    U, B, d, steps = 300, 512, 64, 4
    g = np.random.Generator(np.random.PCG64(9))
    P0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32)
    Q0 = (g.standard_normal((I, d)) * 0.01).astype(np.float32)
    ranges = [parallel.user_range(U, r, world) for r in range(world)]
    streams = [(g.integers(0, hi - lo, steps * B).astype(np.int32), g.integers(0, I, steps * B).astype(np.int32), g.integers(0, I, steps * B).astype(np.int32))
               for lo, hi in ranges]
    lo, hi = ranges[rank]
    st = parallel.DpBprState(sess, P0[lo:hi], Q0, lr=1e-3, reg=0.01, rank=rank, world=world, transport=None)
    u, i, j = (sess.to_dev(x) for x in streams[rank])
    loss = st.train_epoch(u, i, j, B).cpu().numpy()
    st.synchronize()
    gathered = [None] * world
    dist.all_gather_object(gathered, (st.P.cpu().numpy(), st.Q.cpu().numpy(), st.mQ.cpu().numpy(), loss))
    if rank:
        return None
    models = [oracle.BprModel(P0[a:b], Q0, lr=1e-3, reg=0.01) for a, b in ranges]
    ol = []
    for s in range(steps):
        sl = slice(s * B, (s + 1) * B)
        lr_t = None
        dQ, tot = None, 0.0
        for r, m in enumerate(models):  # every rank's local part, summed in rank order (what the all-reduce forms)
            lr_t = float(m.sched.next(1)[0])
            dq_r, l = oracle.bpr_dp_local(m, streams[r][0][sl], streams[r][1][sl], streams[r][2][sl], lr_t)
            dQ = dq_r if dQ is None else dQ + dq_r
            tot += l
        for m in models:
            oracle.adam_dense(m, dQ, lr_t)
        ol.append(tot)
    return {"Q": gathered[0][1], "mQ": gathered[0][2],
            "replicas_identical": all(bool(np.array_equal(x[1], gathered[0][1]) and np.array_equal(x[2], gathered[0][2])) for x in gathered),
            "loss_ok": bool(np.allclose(loss, ol, rtol=1e-5)), "Q_err": float(np.abs(gathered[0][1] - models[0].Q).max()),
            "P_err": float(max(np.abs(x[0] - m.P).max() for x, m in zip(gathered, models)))}


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    try:
        sess = runtime.Session(int(os.environ.get("LOCAL_RANK", "0")))
        parallel.init_peer(sess, rank, world, window_bytes=16 << 20)
        sess.ctx.check(sess.lib.crec_comm_set_timeout(sess.ctx.h, 30.0))
        ex = exchanges(sess, rank, world)
        all_ex = [None] * world
        dist.all_gather_object(all_ex, ex)
        out["exchanges_ok"] = all(all(e) for e in all_ex)
        out["sharded"] = sharded(sess, rank, world)  # exchanges fused with serve / apply (three launches per step)
        sess.ctx.set_option("shard_unfused", 1)
        out["sharded_unfused"] = sharded(sess, rank, world)  # the five-launch form over the same transport
        sess.ctx.set_option("shard_unfused", 0)
        for name, I_ in (("dp", 140), ("dp_two_shot", 300)):
            one = data_parallel(sess, rank, world, I_)  # all-reduce + Adam in one launch
            sess.ctx.set_option("dp_unfused", 1)
            two = data_parallel(sess, rank, world, I_)
            sess.ctx.set_option("dp_unfused", 0)
            if rank == 0:
                one["same_bits_as_unfused"] = bool(np.array_equal(one.pop("Q"), two.pop("Q")) and np.array_equal(one.pop("mQ"), two.pop("mQ")))
                one["unfused"] = two
            out[name] = one
        dist.barrier()
        sess.close()
    finally:
        dist.destroy_process_group()
    if rank == 0:
        print("PEER_RESULT " + json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
