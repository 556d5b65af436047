"""How far the GPU step and the CPU oracle drift apart as Adam steps accumulate (fp32 rounding noise, not a bias)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, oracle
from cleverrec_amd import runtime
from cleverrec_amd.data import RankingData
sess = runtime.Session(0)
def run(U, I, d, B, nr, mode, scale, seed, gseed):
    g = np.random.Generator(np.random.PCG64(gseed))
    deg_hi = max(2, min(I - nr - 1, 20))
    ui_train = {u: [int(x) for x in g.choice(I, size=int(g.integers(1, deg_hi + 1)), replace=False)] for u in range(U) if g.random() < 0.9}
    data = RankingData(U, I, ui_train, {u: [0] for u in list(ui_train)[:3]})
    P0 = (g.standard_normal((U, d)) * scale).astype(np.float32); Q0 = (g.standard_normal((I, d)) * scale).astype(np.float32)
    sess.ctx.seed(seed); u, i, j = sess.ctx.sample_pairwise(data.train_csr, I, nr)
    for steps in sorted({max(1, min(k, len(u) // B)) for k in (8, 64, 512, 1 << 30)}):
        n = steps * B
        st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode=mode)
        om = oracle.BprModel(P0, Q0, lr=1e-3, reg=0.01, adam_mode={"tf1": 0, "lazy": 1}[mode])
        loss = st.train_epoch(sess.to_dev(u[:n]), sess.to_dev(i[:n]), sess.to_dev(j[:n]), B).cpu().numpy()
        ol = oracle.train_model_py(om, steps, u[:n], i[:n], j[:n], B)[1]
        dP = np.abs(st.P.cpu().numpy() - om.P).max(); dQ = np.abs(st.Q.cpu().numpy() - om.Q).max()
        rl = np.abs(loss - ol).max() / max(np.abs(ol).max(), 1e-30)
        print("steps %6d: max|dP| %.2e max|dQ| %.2e rel loss %.2e   (|P| ~ %.2g, Adam step 1e-3)" % (steps, dP, dQ, rl, scale))
run(152, 197, 128, 1, 1, "tf1", 0.01, 1189576617, 1)
run(322, 295, 256, 2048, 1, "lazy", 1.0, 580680542, 2)
run(164, 289, 256, 8192, 4, "tf1", 0.3, 721570491, 3)
