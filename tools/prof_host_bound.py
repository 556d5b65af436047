"""Is the S-big step bound by the host's launch rate?  One epoch call: when the call returns (everything enqueued) against when the GPU
is done.  Run on the GPU box: python tools/prof_host_bound.py [batch] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cleverrec_amd import _lib, runtime
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
users, items, dim = 10_000_000, 1_000_000, 128
sess = runtime.Session(0); dev = sess.device
g = torch.Generator(device=dev).manual_seed(0)
st = runtime.BprState.__new__(runtime.BprState); st.sess = sess
f32 = torch.float32
st.P = torch.empty((users, dim), dtype=f32, device=dev).normal_(0, 0.01, generator=g)
st.Q = torch.empty((items, dim), dtype=f32, device=dev).normal_(0, 0.01, generator=g)
st.mP, st.vP, st.mQ, st.vQ = (torch.zeros_like(t) for t in (st.P, st.P, st.Q, st.Q))
st.tables = _lib.BprTables(*(t.data_ptr() for t in (st.P, st.Q, st.mP, st.vP, st.mQ, st.vQ)), users, items, dim)
st.adam = _lib.Adam(1e-3, 0.9, 0.999, 1e-8, 0.01, _lib.ADAM_LAZY, 0.9, 0.999, 0)
n = steps * batch
u = torch.randint(0, users, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
w = 1.0 / torch.arange(1, items + 1, dtype=torch.float64, device=dev)
cdf = torch.cumsum(w / w.sum(), 0)
i = torch.searchsorted(cdf, torch.rand(n, generator=g, device=dev, dtype=torch.float64)).clamp_(max=items - 1).to(torch.int32)
j = torch.randint(0, items, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
loss = sess.empty((steps,), f32)
st.train_epoch(u, i, j, batch, out=loss); torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    st.train_epoch(u, i, j, batch, out=loss)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("batch %d x %d steps: enqueued after %.2f ms (%.2f us per step), GPU done after %.2f ms (%.2f us per step)"
          % (batch, steps, (t1 - t0) * 1e3, (t1 - t0) / steps * 1e6, (t2 - t0) * 1e3, (t2 - t0) / steps * 1e6))
