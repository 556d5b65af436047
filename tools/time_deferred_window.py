"""Deferred Adam (TF1-exact by replay) on a 4 M x 400 k, d = 128 table: a launch per batch against lazy_window's default (12 per launch
from 16 batches' worth of rows): 134 -> 110 us per step (the replay of ~650 idle steps per touched row is what a step costs here)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cleverrec_amd import runtime, _lib
sess = runtime.Session(0)
U, I, d, B, nb = 4_000_000, 400_000, 128, 6144, 240
dev = sess.device
g = torch.Generator(device=dev).manual_seed(0)
n = nb * B
u = torch.randint(0, U, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
w = 1.0 / torch.arange(1, I + 1, dtype=torch.float64, device=dev)
cdf = torch.cumsum(w / w.sum(), 0)
i = torch.searchsorted(cdf, torch.rand(n, generator=g, device=dev, dtype=torch.float64)).clamp_(max=I - 1).to(torch.int32)
j = torch.randint(0, I, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
for W in (0, -1, 0, -1):
    sess.ctx.set_option("lazy_window", W)
    st = runtime.BprState.__new__(runtime.BprState)
    st.sess = sess
    st.P = torch.empty((U, d), dtype=torch.float32, device=dev).normal_(0, 0.01, generator=g)
    st.Q = torch.empty((I, d), dtype=torch.float32, device=dev).normal_(0, 0.01, generator=g)
    st.mP, st.vP, st.mQ, st.vQ = torch.zeros_like(st.P), torch.zeros_like(st.P), torch.zeros_like(st.Q), torch.zeros_like(st.Q)
    st.tables = _lib.BprTables(*(t.data_ptr() for t in (st.P, st.Q, st.mP, st.vP, st.mQ, st.vQ)), U, I, d)
    st.adam = _lib.Adam(1e-3, 0.9, 0.999, 1e-8, 0.01, _lib.ADAM_DEFERRED, 0.9, 0.999, 0)
    st.lastP = torch.zeros(U, dtype=torch.int32, device=dev); st.lastQ = torch.zeros(I, dtype=torch.int32, device=dev)
    st.deferred = _lib.BprDeferred(st.lastP.data_ptr(), st.lastQ.data_ptr())
    st.train_epoch(u, i, j, B); torch.cuda.synchronize()
    t0 = time.perf_counter(); st.train_epoch(u, i, j, B); torch.cuda.synchronize()
    print("deferred 4M x 400k d=128, lazy_window %d: %.2f us/step" % (W, (time.perf_counter() - t0) / nb * 1e6), flush=True)
    del st
