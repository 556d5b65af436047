"""Per-dispatch HIP-event durations of bpr_step_fused inside 20-batch calls (the driver's `bench.py --steps 20`): the
distribution over many calls of the call's average.  python tools/prof_call_events.py  (CREC_NO_ZERO_COPY=1 to compare)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleverrec_amd import runtime
from cleverrec_amd.data import synthetic

sess = runtime.Session(0)
data = synthetic("S-ml1m", seed=20240601)
U, I, B, d, K = data.user_nums, data.item_nums, 6144, 64, 20
g = np.random.Generator(np.random.PCG64(0))
P0, Q0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32), (g.standard_normal((I, d)) * 0.01).astype(np.float32)
sess.ctx.seed(1)
u, i, j = sess.ctx.sample_pairwise(data.train_csr, I, 4)
du, di, dj = sess.to_dev(u), sess.to_dev(i), sess.to_dev(j)
st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
n = K * B
for _ in range(100):
    st.train_epoch(du[:n], di[:n], dj[:n], B)
torch.cuda.synchronize()
avgs = []
for c in range(200):
    sess.ctx.profile(True)
    st.train_epoch(du[:n], di[:n], dj[:n], B)
    torch.cuda.synchronize()
    pr = sess.ctx.profile_read()
    sess.ctx.profile(False)
    ms, nn = pr["bpr_step_fused"]
    avgs.append(ms / nn * 1e3)
a = np.array(avgs)
print("bpr_step_fused avg per call (us): p10 %.2f p50 %.2f p90 %.2f max %.2f; calls above 7.5 us: %d of %d" % (np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max(), int((a > 7.5).sum()), len(a)))
