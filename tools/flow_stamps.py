"""Phase times of every wave of bpr_steps_flow in ONE step of a 64-step launch (debug build with -DCREC_STAMPS:
STAMP_TOOL=tools/flow_stamps.py tools/wave_stamps.sh [-DCREC_FLOW_STAMP_STEP=30]).  100 MHz wall clock."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleverrec_amd import _lib, runtime  # noqa: E402
from cleverrec_amd.data import synthetic  # noqa: E402

sess = runtime.Session(0)
for k, v in (("flow", 1), ("flow_abl", int(os.environ.get("ABL", "0")))):
    sess.ctx.set_option(k, v)
data = synthetic("S-ml1m", seed=20240601)
U, I, B, d, K = int(os.environ.get("USERS", data.user_nums)), data.item_nums, 6144, 64, 64
g = np.random.Generator(np.random.PCG64(0))
P0, Q0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32), (g.standard_normal((I, d)) * 0.01).astype(np.float32)
sess.ctx.seed(1)
u, i, j = sess.ctx.sample_pairwise(data.train_csr, I, 4)
u = u % U
du, di, dj = sess.to_dev(u), sess.to_dev(i), sess.to_dev(j)
st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
n = K * B
for r in range(5):
    st.train_epoch(du[:n], di[:n], dj[:n], B)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 16384))()
assert sess.lib.crec_debug_stamps(buf, ctypes.c_size_t(8 * 16384)) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(16384, 8).astype(np.int64)
a = a[a[:, 0] > 0][:, :6]
t0 = a[:, 0].min()
top, light, heavy, end = ((a[:, c] - t0) / 100.0 for c in range(4))
recs, nheavy, user = a[:, 5] >> 32, a[:, 5] & 0xff, (a[:, 5] & 0x100) != 0
dur = end - top
print("%d waves stamped; step starts spread over %.2f us (p50 %.2f p90 %.2f), ends over %.2f us" % (len(a), top.max() - top.min(), np.median(top), np.percentile(top, 90), end.max() - end.min()))
for name, sel in (("user waves", user), ("item waves", ~user)):
    if sel.sum() == 0:
        continue
    print("%s (%d): step duration p50 %.2f p90 %.2f max %.2f us | phase A p50 %.2f p90 %.2f max %.2f | barrier + sums p50 %.2f p90 %.2f max %.2f | finish p50 %.2f max %.2f | workgroup's records p50 %d max %d (%d %d)"
          % (name, sel.sum(), *(np.percentile(dur[sel], q) for q in (50, 90, 100)), *(np.percentile((light - top)[sel], q) for q in (50, 90, 100)),
             *(np.percentile((heavy - light)[sel], q) for q in (50, 90, 100)), *(np.percentile((end - heavy)[sel], q) for q in (50, 100)),
             np.median(recs[sel]), recs[sel].max(), np.median(nheavy[sel]), nheavy[sel].max()))
order = np.argsort(-dur)[:12]
print("slowest waves: " + "; ".join("%s recs %d (%d): A %.2f sums %.2f finish %.2f" % ("U" if user[o] else "I", recs[o], nheavy[o], light[o] - top[o], heavy[o] - light[o], end[o] - heavy[o]) for o in order))
# light-walk time against the wave's own record count
for lo, hi in ((0, 17), (17, 33), (33, 65), (65, 97), (97, 129), (129, 100000)):
    sel = (recs >= lo) & (recs < hi)
    if sel.sum():
        print("  workgroup's records [%d, %d): %d waves, step p50 %.2f max %.2f us, phase A p50 %.2f" % (lo, hi, sel.sum(), np.median(dur[sel]), dur[sel].max(), np.median((light - top)[sel])))
