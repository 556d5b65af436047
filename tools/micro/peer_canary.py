"""Peer transport (csrc/crec_peer.hip) check + micro-benchmark, one process per rank.

  python tools/micro/peer_canary.py --spawn N [--device-per-rank]   start N ranks from here (all on cuda:0 unless --device-per-rank)
  RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set     be one rank (bench.py starts it this way before its N > 1 run)

Every rank maps every other rank's window (HIP IPC), then
  * correctness: ragged all-to-all-v rounds and all-reduces of 1 .. 237 184 floats, bit for bit against what the ranks say they sent
    (rank-order sums), back to back so both window halves and both flag parities are reused;
  * timing: the all-reduce of the replicated-Q step (3706 x 64 floats) and the all-to-all-v of a row-sharded step's rows
    (9 600 rows x 512 B per rank, spread evenly over the peers), enqueued back to back on one stream, per-call microseconds.
Rank 0 prints one JSON line {"ok": ..}; exit code 0 only if every rank saw every exchange arrive intact.  gloo carries the handles."""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")


def spawn(n, device_per_rank):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r if device_per_rank else 0), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    rc = 0
    for p in procs:
        rc = p.wait() or rc
    return rc


def rank_main():
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from cleverrec_amd import parallel, runtime
    rank, world, local = (int(os.environ[k]) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"))
    os.environ.pop("TORCHELASTIC_USE_AGENT_STORE", None)  # (a canary started from a torchrun worker: this group's rank 0 hosts the store)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok, res = True, {"world": world}
    try:
        sess = runtime.Session(local)
        parallel.init_peer(sess, rank, world, window_bytes=64 << 20)
        sess.ctx.check(sess.lib.crec_comm_set_timeout(sess.ctx.h, 20.0))
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731

        def a2a_args(cnt):
            scnt, rcnt = cnt[rank].astype(np.int64), cnt[:, rank].astype(np.int64)
            sdsp = np.concatenate([[0], np.cumsum(scnt)[:-1]]).astype(np.int64)
            rdsp = np.concatenate([[0], np.cumsum(rcnt)[:-1]]).astype(np.int64)
            return scnt, sdsp, rcnt, rdsp

        def a2a(d_send, d_recv, args):
            scnt, sdsp, rcnt, rdsp = args
            sess.ctx.check(sess.lib.crec_debug_alltoallv(sess.ctx.h, sess.ptr(d_send), p(scnt), p(sdsp), sess.ptr(d_recv), p(rcnt), p(rdsp), sess.stream_ptr()))

        def words(r, src, dst, n):
            return np.random.Generator(np.random.PCG64(77 * r + 13 * src + dst)).integers(0, 2 ** 31, int(n)).astype(np.int32)

        for r in range(8):
            cnt = np.random.Generator(np.random.PCG64(1000 + r)).integers(0, 5000, (world, world)) * (4 if r % 2 else 1)
            cnt[np.random.Generator(np.random.PCG64(2000 + r)).random((world, world)) < 0.2] = 0
            send = np.concatenate([words(r, rank, q, cnt[rank, q]) for q in range(world)] + [np.zeros(1, np.int32)])
            d_recv = sess.zeros(int(cnt[:, rank].sum()) + 1, torch.int32)
            a2a(sess.to_dev(send), d_recv, a2a_args(cnt))
            want = np.concatenate([words(r, q, rank, cnt[q, rank]) for q in range(world)] + [np.zeros(1, np.int32)])
            ok = ok and bool(np.array_equal(d_recv.cpu().numpy(), want))
            n = [1, 37, 4096, 237184][r % 4]
            d_x = sess.to_dev(np.random.Generator(np.random.PCG64(5 * r + rank)).standard_normal(n).astype(np.float32))
            sess.ctx.check(sess.lib.crec_debug_allreduce(sess.ctx.h, sess.ptr(d_x), n, sess.stream_ptr()))
            acc = np.zeros(n, np.float32)
            for q in range(world):
                acc = acc + np.random.Generator(np.random.PCG64(5 * r + q)).standard_normal(n).astype(np.float32)
            ok = ok and bool(np.array_equal(d_x.cpu().numpy(), acc))
        sess.ctx.check(sess.lib.crec_comm_status(sess.ctx.h, sess.stream_ptr()))

        def timed(fn, reps=50, batches=5):
            """per-call microseconds: [device time between two events around `reps` back-to-back calls, host wall time], the best of
            `batches` such runs (a young process has one ~40 ms hiccup in its first second or two), max over ranks"""
            fn()
            best = None
            for _ in range(batches):
                torch.cuda.synchronize()
                dist.barrier()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                t = torch.tensor([e0.elapsed_time(e1) * 1e-3, dt], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                best = t if best is None or float(t[0]) < float(best[0]) else best
            return [round(float(x) / reps * 1e6, 2) for x in best]

        n = 3706 * 64
        d_x = sess.zeros(n, torch.float32)
        res["allreduce_3706x64_f32_us"] = timed(lambda: sess.ctx.check(sess.lib.crec_debug_allreduce(sess.ctx.h, sess.ptr(d_x), n, sess.stream_ptr())))
        d_1 = sess.zeros(64, torch.float32)
        res["allreduce_1_float_us"] = timed(lambda: sess.ctx.check(sess.lib.crec_debug_allreduce(sess.ctx.h, sess.ptr(d_1), 1, sess.stream_ptr())))
        rows = 9600 // world * world
        cnt = np.full((world, world), rows // world * 128, np.int64)  # 512-byte rows, in 4-byte elements
        d_s, d_r = sess.zeros(rows * 128, torch.int32), sess.zeros(rows * 128, torch.int32)
        args = a2a_args(cnt)
        res["alltoallv_9600_rows_512B_us"] = timed(lambda: a2a(d_s, d_r, args))
        res["us_are"] = "[device, host wall] per call, best of 5 runs of 50 calls back to back on one stream, max over ranks; the host side is a ctypes call per exchange (about 6 us, 11 us with count arrays): a device figure at that level is the host's"
        sess.ctx.check(sess.lib.crec_comm_status(sess.ctx.h, sess.stream_ptr()))
        res["devices"] = "one per rank" if local == rank and world > 1 and torch.cuda.device_count() >= world else "cuda:%d" % local
        sess.close()
    except Exception as e:  # noqa: BLE001
        ok = False
        res["error"] = repr(e)
    try:
        oks = [None] * world
        dist.all_gather_object(oks, ok)
        ok = all(oks)
    finally:
        dist.destroy_process_group()
    res["ok"] = ok
    if rank == 0:
        print(json.dumps(res), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--spawn", type=int, default=0)
    ap.add_argument("--device-per-rank", action="store_true")
    a = ap.parse_args()
    sys.exit(spawn(a.spawn, a.device_per_rank) if a.spawn else rank_main())
