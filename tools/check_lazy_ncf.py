"""NeuMF steps, TF1 Adam and LazyAdam side by side on the same draws: worst |delta| / envelope per tensor and seed
(is an envelope overshoot the lazy rule's or the draw's?).  Run on the GPU box."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _adam_close import assert_adam_close
from cleverrec_amd import runtime
from oracle import models_torch as MT
import test_gpu_ncf as T

sess = runtime.Session(0)
U, I, B, d, layers = 300, 200, 128, 16, [32, 16]
for seed in range(41, 47):
    for opt in ("Adam", "LazyAdam"):
        g = np.random.Generator(np.random.PCG64(seed))
        u, i, y = T._instances(seed - 32, U, I, 3 * B + 17)
        Pg, Qg, Pm, Qm = T._rand(g, U, d), T._rand(g, I, d), T._rand(g, U, layers[0] // 2), T._rand(g, I, layers[0] // 2)
        Ws, bs, last = T._tower(g, layers[0], len(layers))
        h = T._rand(g, d + last, scale=0.3)
        st = runtime.NcfState(sess, "NeuMF", {"P_gmf": Pg, "Q_gmf": Qg, "P_mlp": Pm, "Q_mlp": Qm}, Ws, bs, h, lr=1e-3, reg_gmf=1e-2, reg_mlp=1e-3, optimizer=opt)
        om = MT.NeuMF(Pg, Qg, Pm, Qm, Ws, bs, h, lr=1e-3, reg1=1e-2, reg2=1e-3, optimizer=opt)
        loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(y), B).cpu().numpy()
        for b in range(len(loss)):
            om.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], y[b * B:(b + 1) * B])
        worst = {}
        for k in ("P_gmf", "Q_gmf", "P_mlp", "Q_mlp"):
            rep = []
            try:
                assert_adam_close(om, k, st.t[k], *st.slots[k], report=rep, scale=1000.0)
            except AssertionError as e:
                print("  ", e)
            worst[k] = max(r[2] for r in rep) * 1000.0
        print(seed, opt, {k: round(v, 2) for k, v in worst.items()})
