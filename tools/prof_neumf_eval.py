import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from cleverrec_amd import runtime
from cleverrec_amd.data import synthetic
sess = runtime.Session(0)
data = synthetic("S-ml1m"); U, I = data.user_nums, data.item_nums
g = np.random.Generator(np.random.PCG64(0)); rnd = lambda *s: (g.standard_normal(s) * 0.05).astype(np.float32)
Ws, bs, w = [], [], 128
for _ in range(3):
    Ws.append(rnd(w, w // 2)); bs.append(rnd(w // 2)); w //= 2
st = runtime.NcfState(sess, "NeuMF", {"P_gmf": rnd(U, 64), "Q_gmf": rnd(I, 64), "P_mlp": rnd(U, 64), "Q_mlp": rnd(I, 64)}, Ws, bs, rnd(80), lr=1e-3, reg_gmf=1e-2, reg_mlp=1e-3)
users = sess.to_dev(data.test_users); sp, si = sess.to_dev(data.seen_csr[0]), sess.to_dev(data.seen_csr[1])
def ev():
    for t in range(0, len(users), 1024):
        cur = users[t:t + 1024]
        sc = runtime.ncf_scores(sess, st, cur)
        runtime.topk_scores(sess, sc, cur, sp, si, 20)
ev(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): ev()
torch.cuda.synchronize(); print("NeuMF eval ms", (time.perf_counter() - t0) / 3 * 1e3, "users/s", len(users) / ((time.perf_counter() - t0) / 3))
