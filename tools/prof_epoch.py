"""Where an epoch's wall time goes with the sampler prefetch on / off (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cleverrec_amd import runtime
from cleverrec_amd.data import synthetic
from cleverrec_amd.utils import sampler as S
sess = runtime.Session(0)
data = synthetic("S-ml1m")
P0 = (np.random.default_rng(0).standard_normal((data.user_nums, 64)) * 0.01).astype(np.float32)
Q0 = (np.random.default_rng(1).standard_normal((data.item_nums, 64)) * 0.01).astype(np.float32)
st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
for prefetch in (False, True, False, True):
    S.drop_prefetch(); np.random.seed(1)
    rows = []
    for e in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        nb, u, i, j = S.pairwise_ranking_sampler(data, 4, 6144, prefetch_next=prefetch); t1 = time.perf_counter()
        du, di, dj = sess.to_dev(u), sess.to_dev(i), sess.to_dev(j); t2 = time.perf_counter()
        st.train_epoch(du, di, dj, 6144); t3 = time.perf_counter()
        torch.cuda.synchronize(); t4 = time.perf_counter()
        rows.append([round((b - a) * 1e3, 1) for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t0, t4))])
    print("prefetch", prefetch, "[sample, upload, enqueue, drain, total] ms per epoch:", rows)
S.drop_prefetch()
