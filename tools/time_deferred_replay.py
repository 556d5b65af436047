"""Deferred Adam where every touched row has sat out ~700 steps (the steady state of a 4 M x 400 k table at 6144 triples per batch):
us per step of 120 batches.  The state is made up: every row's stamp says step 1, the optimizer stands at step 700."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cleverrec_amd import runtime, _lib
sess = runtime.Session(0)
U, I, d, B, nb = 4_000_000, 400_000, 128, 6144, 120
dev = sess.device
g = torch.Generator(device=dev).manual_seed(0)
n = nb * B
u = torch.randint(0, U, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
i = torch.randint(0, I, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
j = torch.randint(0, I, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
for W in (0, -1):
    sess.ctx.set_option("lazy_window", W)
    st = runtime.BprState.__new__(runtime.BprState)
    st.sess = sess
    st.P = torch.empty((U, d), dtype=torch.float32, device=dev).normal_(0, 0.01, generator=g)
    st.Q = torch.empty((I, d), dtype=torch.float32, device=dev).normal_(0, 0.01, generator=g)
    st.mP, st.mQ = torch.empty_like(st.P).normal_(0, 1e-3, generator=g), torch.empty_like(st.Q).normal_(0, 1e-3, generator=g)
    st.vP, st.vQ = torch.full_like(st.P, 1e-6), torch.full_like(st.Q, 1e-6)
    st.tables = _lib.BprTables(*(t.data_ptr() for t in (st.P, st.Q, st.mP, st.vP, st.mQ, st.vQ)), U, I, d)
    s0 = 700
    st.adam = _lib.Adam(1e-3, 0.9, 0.999, 1e-8, 0.01, _lib.ADAM_DEFERRED, 0.9 ** (s0 + 1), 0.999 ** (s0 + 1), s0)
    st.lastP = torch.ones(U, dtype=torch.int32, device=dev); st.lastQ = torch.ones(I, dtype=torch.int32, device=dev)
    st.deferred = _lib.BprDeferred(st.lastP.data_ptr(), st.lastQ.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter(); st.train_epoch(u, i, j, B); torch.cuda.synchronize()
    print("deferred, rows idle ~700 steps, lazy_window %d: %.2f us/step" % (W, (time.perf_counter() - t0) / nb * 1e6), flush=True)
    del st
