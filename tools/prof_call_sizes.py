"""Per-kernel HIP-event times of crec_bpr_train_epoch calls of 1 ... 455 batches (tf1, S-ml1m): the index kernel is one
workgroup per batch and side, so a call of 1 to 64 batches pays the same ~50 us for it.  python tools/prof_call_sizes.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleverrec_amd import runtime
from cleverrec_amd.data import synthetic
sess = runtime.Session(0)
data = synthetic("S-ml1m", seed=20240601)
U, I, B, d = data.user_nums, data.item_nums, 6144, 64
g = np.random.Generator(np.random.PCG64(0))
P0, Q0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32), (g.standard_normal((I, d)) * 0.01).astype(np.float32)
sess.ctx.seed(1)
u, i, j = sess.ctx.sample_pairwise(data.train_csr, I, 4)
du, di, dj = sess.to_dev(u), sess.to_dev(i), sess.to_dev(j)
st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
for K in (1, 2, 5, 20, 64, 128, 455):
    n = K * B
    for _ in range(30):
        st.train_epoch(du[:n], di[:n], dj[:n], B)
    torch.cuda.synchronize()
    sess.ctx.profile(True)
    for _ in range(50):
        st.train_epoch(du[:n], di[:n], dj[:n], B)
        torch.cuda.synchronize()
    p = sess.ctx.profile_read()
    sess.ctx.profile(False)
    print(K, {k: round(v[0] / max(v[1], 1) * 1e3, 2) for k, v in p.items() if v[1]})
