"""One NCF shape under the rounding envelope of oracle/models_torch.py (tests/_adam_close.py): the principled form of the
check behind a flag of tests/soak/soak_models.py.  python tools/check_ncf_shape.py MLP 111 12 1000 2500 64 128 3 Adam"""
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from oracle import models_torch as MT
from cleverrec_amd import runtime
from _adam_close import assert_adam_close

kind, U, I, B, N, dg, L0, nl, optim = sys.argv[1], *map(int, sys.argv[2:9]), sys.argv[9]
sess = runtime.Session(0)
worst = 0.0
for seed in range(20):
    g = np.random.Generator(np.random.PCG64(seed))
    rnd = lambda *s, scale=0.1: (g.standard_normal(s) * scale).astype(np.float32)
    u, i = g.integers(0, U, N).astype(np.int32), g.integers(0, I, N).astype(np.int32)
    y = (g.random(N) < 0.2).astype(np.float32)
    Ws, bs, w = [], [], L0
    for _ in range(nl):
        Ws.append(rnd(w, w // 2, scale=0.2)); bs.append(rnd(w // 2, scale=0.05)); w //= 2
    if kind == "MLP":
        P, Q, h = rnd(U, L0 // 2), rnd(I, L0 // 2), rnd(w, scale=0.5)
        st = runtime.NcfState(sess, "MLP", {"P_mlp": P, "Q_mlp": Q}, Ws, bs, h, lr=1e-3, reg_mlp=0.01, optimizer=optim)
        om = MT.MLP(P, Q, Ws, bs, h, lr=1e-3, reg=0.01, optimizer=optim)
        names = [("P", st.t["P_mlp"], st.slots["P_mlp"]), ("Q", st.t["Q_mlp"], st.slots["Q_mlp"])]
    else:
        Pg, Qg, Pm, Qm, h = rnd(U, dg), rnd(I, dg), rnd(U, L0 // 2), rnd(I, L0 // 2), rnd(dg + w, scale=0.3)
        st = runtime.NcfState(sess, "NeuMF", {"P_gmf": Pg, "Q_gmf": Qg, "P_mlp": Pm, "Q_mlp": Qm}, Ws, bs, h, lr=1e-3, reg_gmf=1e-2, reg_mlp=1e-3, optimizer=optim)
        om = MT.NeuMF(Pg, Qg, Pm, Qm, Ws, bs, h, lr=1e-3, reg1=1e-2, reg2=1e-3, optimizer=optim)
        names = [(k, st.t[k], st.slots[k]) for k in ("P_gmf", "Q_gmf", "P_mlp", "Q_mlp")]
    loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(y), B).cpu().numpy()
    ol = [om.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], y[b * B:(b + 1) * B]) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=2e-5, atol=1e-6)
    for name, dev, (m, v) in names:
        rep = []
        if optim == "Adam":
            assert_adam_close(om, name, dev, m, v, report=rep)
            worst = max(worst, max(r[2] for r in rep))
        ref = om.numpy(name)
        d = np.abs(dev.cpu().numpy() - ref)
        print("seed %2d %-6s max |diff| %.3g, elements beyond 2e-5: %d of %d" % (seed, name, d.max(), int((d > 2e-5).sum()), d.size))
print("all within the envelope; worst ratio of bound %.3f" % worst)
