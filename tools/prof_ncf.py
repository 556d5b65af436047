import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from cleverrec_amd import runtime
from cleverrec_amd.data import synthetic
sess = runtime.Session(0)
data = synthetic("S-ml1m"); U, I, B = data.user_nums, data.item_nums, 6144
g = np.random.Generator(np.random.PCG64(0)); rnd = lambda *s: (g.standard_normal(s) * 0.05).astype(np.float32)
sess.ctx.seed(1); u, i, y = sess.ctx.sample_pointwise(data.train_csr, I, 4)
n = 100 * B
du, di, dy = sess.to_dev(u[:n]), sess.to_dev(i[:n]), sess.to_dev(y[:n])
Ws, bs, w = [], [], 128
for _ in range(3):
    Ws.append(rnd(w, w // 2)); bs.append(rnd(w // 2)); w //= 2
for kind, tabs, WW, BB, h in (("NeuMF", {"P_gmf": rnd(U, 64), "Q_gmf": rnd(I, 64), "P_mlp": rnd(U, 64), "Q_mlp": rnd(I, 64)}, Ws, bs, rnd(80)),
                              ("GMF", {"P_gmf": rnd(U, 64), "Q_gmf": rnd(I, 64)}, [], [], rnd(64))):
    st = runtime.NcfState(sess, kind, tabs, WW, BB, h, lr=1e-3, reg_gmf=1e-2, reg_mlp=1e-3)
    st.train_epoch(du, di, dy, B); torch.cuda.synchronize()
    sess.ctx.profile(True); st.train_epoch(du, di, dy, B); torch.cuda.synchronize()
    print(kind, {k: (round(ms / nn * 1e3, 1), nn) for k, (ms, nn) in sess.ctx.profile_read().items() if nn}); sess.ctx.profile(False)

# ---- FM (C4) and CML, same per-kernel brackets
n = 100 * B
xu, xi = g.integers(0, U, n), g.integers(0, I, n)
x_idx = np.stack([xu, U + xi], 1).astype(np.int32)
fm = runtime.FmState(sess, np.zeros(1, np.float32), rnd(U + I + 1), rnd(U + I + 1, 32), lr=1e-3, reg=1e-3)
dxi, dxv, dyy = sess.to_dev(x_idx), sess.to_dev(np.ones((n, 2), np.float32)), sess.to_dev(g.integers(1, 6, n).astype(np.float32))
sess.ctx.seed(1); cu, ci, cn = sess.ctx.sample_cml(data.train_csr, I, 20)
cml = runtime.CmlState(sess, rnd(U, 128), rnd(I, 128), lr=1e-3, reg=10.0, margin=1.0)
dcu, dci, dcn = sess.to_dev(cu[:n]), sess.to_dev(ci[:n]), sess.to_dev(cn[:n])
import time
for kind, fn in (("FM", lambda: fm.train_epoch(dxi, dxv, dyy, B)), ("CML", lambda: cml.train_epoch(dcu, dci, dcn, B))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    sess.ctx.profile(True); fn(); torch.cuda.synchronize()
    print(kind, "us/step %.1f" % (dt / 100 * 1e6), {k: (round(ms / nn * 1e3, 1), nn) for k, (ms, nn) in sess.ctx.profile_read().items() if nn}); sess.ctx.profile(False)
