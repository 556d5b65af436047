#!/usr/bin/env python3
"""TF1-exact Adam at growing catalogues: the swept step (every row every step: bpr_steps_flow where the rows fit its workgroups,
bpr_step_fused beyond) against deferred Adam + the flush at the end of the call (the same results, touched rows only).
us per step over calls of 200 batches of 6144 triples, users uniform, positives Zipf(1.0)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cleverrec_amd import runtime  # noqa: E402

sess = runtime.Session(0)
B, nb = 6144, 200
for U, I, d in ((6040, 3706, 64), (20000, 10000, 64), (50000, 20000, 64), (138000, 27000, 64), (138000, 27000, 128), (500000, 100000, 64)):
    g = np.random.Generator(np.random.PCG64(1))
    n = nb * B
    u = g.integers(0, U, n).astype(np.int32)
    w = 1.0 / np.arange(1, I + 1)
    i = g.choice(I, n, p=w / w.sum()).astype(np.int32)
    j = g.integers(0, I, n).astype(np.int32)
    P0, Q0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32), (g.standard_normal((I, d)) * 0.01).astype(np.float32)
    du, di, dj = (sess.to_dev(x) for x in (u, i, j))
    row = []
    for mode in ("tf1", "deferred"):
        st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode=mode)
        for _ in range(2):
            st.train_epoch(du, di, dj, B)
            st.flush()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            st.train_epoch(du, di, dj, B)
            st.flush()
        torch.cuda.synchronize()
        row.append("%s %.2f" % (mode, (time.perf_counter() - t0) / (3 * nb) * 1e6))
        del st
    print("%7d x %6d d=%3d: us/step  %s" % (U, I, d, "   ".join(row)), flush=True)
