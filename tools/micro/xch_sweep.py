"""One rank's exchange kernel (peer transport, world of one: the launch, its flag hand-shake with itself, copy-in and pull on the same
GPU) over payload sizes: microseconds per call, back to back on one stream, and the slowest single calls.  python tools/micro/xch_sweep.py"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from cleverrec_amd import parallel, runtime
sess = runtime.Session(0)
blob = (C.c_char * 192)()
sess.ctx.check(sess.lib.crec_comm_peer_create(sess.ctx.h, 64 << 20, blob))
sess.ctx.check(sess.lib.crec_comm_peer_connect(sess.ctx.h, 0, 1, C.c_char_p(bytes(blob))))
p = lambda a: a.ctypes.data_as(C.c_void_p)
def bench(fn, reps=100):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    each = []
    for _ in range(30):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); each.append((time.perf_counter() - t0) * 1e6)
    return round(e0.elapsed_time(e1) * 1e3 / reps, 2), [round(x) for x in sorted(each)[-3:]]
for order in ("a2a-first", "allreduce-first"):
    for rows in (1, 75, 600, 9600, 19200):
        n = rows * 128
        d_s, d_r, x = sess.zeros(n, torch.int32), sess.zeros(n, torch.int32), sess.zeros(n, torch.float32)
        scnt = np.array([n], np.int64); z = np.array([0], np.int64)
        fa = lambda: sess.ctx.check(sess.lib.crec_debug_alltoallv(sess.ctx.h, sess.ptr(d_s), p(scnt), p(z), sess.ptr(d_r), p(scnt), p(z), sess.stream_ptr()))
        fr = lambda: sess.ctx.check(sess.lib.crec_debug_allreduce(sess.ctx.h, sess.ptr(x), n, sess.stream_ptr()))
        for name, fn in ((("a2a", fa), ("allreduce", fr)) if order == "a2a-first" else (("allreduce", fr), ("a2a", fa))):
            print(order, rows, "rows", n * 4, "bytes", name, *bench(fn), flush=True)
