"""Two ranks of the peer transport as THREADS of one process on one GPU, the row-sharded epoch, again and again (fresh sessions each
time): which outcomes occur with the ranks' streams at different priorities (`prio`) and at the same one (`same`).  What it showed:
same-priority streams may share a hardware queue -- a rank's kernel that waits in the GPU for its peer then starves it (symmetric
time-outs, about one run in four); and (tests/test_gpu_peer.py) a rank that copies from pageable host memory while its peer's kernel
already waits drains that kernel's stream first.  Both are properties of ranks-as-threads; one process per GPU has neither.
usage (GPU box): python tools/micro/thread_ranks_stress.py prio|same [iterations]"""
import os, sys, threading, time
import numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from cleverrec_amd import parallel, runtime
G = 2
U, I, B, d = 400, 301, 256, 64
mode = sys.argv[1] if len(sys.argv) > 1 else "prio"
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    g = np.random.Generator(np.random.PCG64(3))
    P0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32)
    Q0 = (g.standard_normal((I, d)) * 0.01).astype(np.float32)
    ranges = [parallel.user_range(U, r, G) for r in range(G)]
    streams = []
    for r, (lo, hi) in enumerate(ranges):
        n = 2 * B + [40, 7][r]
        streams.append((g.integers(0, hi - lo, n).astype(np.int32), np.minimum(g.zipf(1.3, n) - 1, I - 1).astype(np.int32), g.integers(0, I, n).astype(np.int32)))
    sessions = [runtime.Session(0) for _ in range(G)]
    parallel.init_peer_local(sessions, window_bytes=8 << 20)
    for s in sessions:
        s.ctx.check(s.lib.crec_comm_set_timeout(s.ctx.h, 3.0))
    rs = [torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1 if mode == "prio" else 0)]
    errors, done = {}, {}
    def work(r):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(rs[r]):
                s = sessions[r]
                lo, hi = ranges[r]
                st = parallel.ShardedBprState(s, P0[lo:hi], parallel.shard_items(Q0, r, G), I, 1e-3, 0.01, r, G)
                u, i, j = (s.to_dev(x) for x in streams[r])
                t0 = time.time()
                loss = st.train_epoch(u, i, j, B).cpu().numpy()
                done[r] = time.time() - t0
        except Exception as e:
            errors[r] = str(e)[:160]
    th = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    [t.start() for t in th]; [t.join(60) for t in th]
    print("iter %d (%s): done %s errors %s" % (it, mode, {k: round(v, 3) for k, v in done.items()}, errors), flush=True)
    for s in sessions:
        s.close()
    # churn: what earlier tests of a suite leave behind -- a few more streams
    _ = [torch.cuda.Stream() for _ in range(3)]
