"""An epoch of the headline workload with its sampling (runtime.MtSampler: the next epoch drawn on a side stream beside this epoch's
steps): does it matter which priority the side stream has?  Prints the steady epoch time with the side stream at the default
priority and at the device's lowest.  Run on the GPU box."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cleverrec_amd import runtime
from cleverrec_amd.data import synthetic

sess = runtime.Session(0)
data = synthetic("S-ml1m", seed=20240601)
B, NEG, D = 6144, 4, 64
g = np.random.Generator(np.random.PCG64(0))
P0 = (g.standard_normal((data.user_nums, D)) * 0.01).astype(np.float32)
Q0 = (g.standard_normal((data.item_nums, D)) * 0.01).astype(np.float32)
st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
hip = C.CDLL("libamdhip64.so")
lo, hi = C.c_int(), C.c_int()
hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
print("priority range: least %d greatest %d" % (lo.value, hi.value))

def run(side):
    ms = runtime.MtSampler(sess, data)
    if side is not None:
        ms.side = side
    np.random.seed(1)
    t = []
    for e in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cu, ci, cj = ms.next_epoch("pairwise", NEG, ahead=True)
        st.train_epoch(cu, ci, cj, B)
        torch.cuda.synchronize()
        t.append(time.perf_counter() - t0)
    ms.drop(); ms.close()
    tt = sorted(t[2:])
    return tt[len(tt) // 2] * 1e3, tt[0] * 1e3, int(cu.numel())

for name, prio in (("default", None), ("lowest", lo.value), ("default", None), ("lowest", lo.value)):
    side = None
    if prio is not None:
        h = C.c_void_p()
        assert hip.hipStreamCreateWithPriority(C.byref(h), C.c_uint(1), C.c_int(prio)) == 0  # hipStreamNonBlocking
        side = torch.cuda.ExternalStream(h.value, device=sess.device)
    med, best, n = run(side)
    print("side stream %-8s: epoch median %.2f ms (best %.2f) = %.0f M triples/s" % (name, med, best, n / med / 1e3))
