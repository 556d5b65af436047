"""Where the NeuMF full-catalogue comparison (tests/test_gpu_ncf.py) leaves its envelope: rows, their occurrences per batch."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cleverrec_amd import runtime
from cleverrec_amd.data import synthetic
import test_gpu_ncf as T
MT = T.MT
sess = runtime.Session(0)
data = synthetic("S-ml1m")
U, I, B, d = data.user_nums, data.item_nums, 6144, 64
g = np.random.Generator(np.random.PCG64(31))
Pg, Qg, Pm, Qm = T._rand(g, U, d), T._rand(g, I, d), T._rand(g, U, 64), T._rand(g, I, 64)
Ws, bs, last = T._tower(g, 128, 3)
h = T._rand(g, d + last, scale=0.3)
sess.ctx.seed(3)
u, i, y = sess.ctx.sample_pointwise(data.train_csr, I, 4)
n = 3 * B + 1234
u, i, y = u[:n], i[:n], y[:n]
print("instances checksum", int(u.astype(np.int64).sum()), int(i.astype(np.int64).sum()), float(y.sum()))
st = runtime.NcfState(sess, "NeuMF", {"P_gmf": Pg, "Q_gmf": Qg, "P_mlp": Pm, "Q_mlp": Qm}, Ws, bs, h, lr=1e-3, reg_gmf=1e-2, reg_mlp=1e-3)
om = MT.NeuMF(Pg, Qg, Pm, Qm, Ws, bs, h, lr=1e-3, reg1=1e-2, reg2=1e-3)
loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(y), B).cpu().numpy()
ol = [om.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], y[b * B:(b + 1) * B]) for b in range(len(loss))]
print("loss", loss, ol)
for name, ids in (("P_mlp", u), ("Q_mlp", i), ("P_gmf", u)):
    m, v = T._slots(st, st.t[name])
    tol_var, tol_m, tol_v = om.bounds(name)
    mr = om.opt.m[name].numpy().astype(np.float64)
    dm = np.abs(m.cpu().numpy().astype(np.float64) - mr)
    bound = tol_m + 1e-6 * np.abs(mr) + 1e-10
    ratio = dm / bound
    rows = np.unique(np.nonzero(ratio > 2)[0])
    print(name, "m: elements beyond 2 envelopes", int((ratio > 2).sum()), "in", len(rows), "rows; worst", ratio.max())
    for r in rows[:12]:
        occ = [int((ids[b * B:(b + 1) * B] == r).sum()) for b in range(4)]
        print("   row", r, "occ per batch", occ, "elements >2:", int((ratio[r] > 2).sum()), "max ratio %.2f" % ratio[r].max(), "max |dm| %.3g" % dm[r].max(), "|m| max %.3g" % np.abs(mr[r]).max())
