"""Where a 20-batch crec_bpr_train_epoch call spends its host time (the driver's `bench.py --steps 20`): the Python wrapper
(slices, output allocation, ctypes marshalling) against the C call alone.  python tools/prof_call.py"""
import ctypes as C
import os
import sys
import time

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


def wrapped():
    st.train_epoch(du[:n], di[:n], dj[:n], B)


su, si, sj = du[:n], di[:n], dj[:n]
loss = sess.empty((K,), torch.float32)
pu, pi, pj, pl, sp = sess.ptr(su), sess.ptr(si), sess.ptr(sj), sess.ptr(loss), sess.stream_ptr()
tb, ad = C.byref(st.tables), C.byref(st.adam)


def bare():
    sess.lib.crec_bpr_train_epoch(sess.ctx.h, tb, ad, pu, pi, pj, n, B, pl, sp)


for name, fn in (("python wrapper", wrapped), ("C call alone", bare), ("python wrapper", wrapped), ("C call alone", bare)):
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(300):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    a = np.array(ts) * 1e6
    print("%-15s call returns after %.1f us (median), done after %.1f us -> %.0f M triples/s" % (name, np.median(a[:, 0]), np.median(a[:, 1]), n / np.median(a[:, 1])))
