"""How the host learns that a 20-batch call is done: torch.cuda.synchronize() alone against polling an event first, and
the same under ROC_ACTIVE_WAIT_TIMEOUT (set in the environment before starting).  python tools/prof_sync.py"""
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
su, si, sj = du[:n], di[:n], dj[:n]
loss = sess.empty((K,), torch.float32)
pu, pi, pj, pl, sp = sess.ptr(su), sess.ptr(si), sess.ptr(sj), sess.ptr(loss), sess.stream_ptr()
tb, ad = C.byref(st.tables), C.byref(st.adam)
stream = sess.stream if hasattr(sess, "stream") else torch.cuda.current_stream()


def bare():
    sess.lib.crec_bpr_train_epoch(sess.ctx.h, tb, ad, pu, pi, pj, n, B, pl, sp)


def wait_sync():
    torch.cuda.synchronize()


ev = torch.cuda.Event()


def wait_poll():
    ev.record(stream)
    while not ev.query():
        pass
    torch.cuda.synchronize()


def wait_stream():
    stream.synchronize()


print("ROC_ACTIVE_WAIT_TIMEOUT =", os.environ.get("ROC_ACTIVE_WAIT_TIMEOUT"))
for name, wait in (("synchronize", wait_sync), ("event poll", wait_poll), ("stream sync", wait_stream), ("synchronize", wait_sync), ("event poll", wait_poll)):
    for _ in range(200):
        bare()
    torch.cuda.synchronize()
    ts = []
    for _ in range(300):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bare()
        t1 = time.perf_counter()
        wait()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    a = np.array(ts) * 1e6
    print("%-12s call returns after %.1f us (median), done after %.1f us -> %.0f M triples/s" % (name, np.median(a[:, 0]), np.median(a[:, 1]), n / np.median(a[:, 1])))
# an empty stream: what a synchronize costs with nothing to wait for, and right after one tiny kernel
x = torch.zeros(64, device="cuda")
for name, wait in (("synchronize", wait_sync), ("event poll", wait_poll)):
    ts = []
    for _ in range(300):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x.add_(1.0)
        wait()
        ts.append(time.perf_counter() - t0)
    print("one tiny kernel + %-12s %.1f us (median)" % (name, np.median(ts) * 1e6))
