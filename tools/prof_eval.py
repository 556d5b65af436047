import sys, os, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from cleverrec_amd import runtime
from cleverrec_amd.data import synthetic
sess = runtime.Session(0)
data = synthetic("S-ml1m")
g = np.random.Generator(np.random.PCG64(0))
P = sess.to_dev((g.standard_normal((data.user_nums, 64)) * 0.3).astype(np.float32)); Q = sess.to_dev((g.standard_normal((data.item_nums, 64)) * 0.3).astype(np.float32))
users = sess.to_dev(data.test_users); sp, si = sess.to_dev(data.seen_csr[0]), sess.to_dev(data.seen_csr[1]); tp, ti = sess.to_dev(data.test_csr[0]), sess.to_dev(data.test_csr[1])
def once():
    ids, _ = runtime.eval_topk(sess, P, Q, users, sp, si, 20)
    return runtime.rank_metrics(sess, ids, users, [10, 20], tp, ti)
for shape in ("4", "8", "4", "8"):
    sess.ctx.set_option("eval_shape", int(shape))
    once(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): once()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    sess.ctx.profile(True); once(); once(); torch.cuda.synchronize()
    pr = {k: round(ms / n * 1e3, 1) for k, (ms, n) in sess.ctx.profile_read().items() if n}; sess.ctx.profile(False)
    print("shape", shape, "ms %.3f" % (dt * 1e3), pr)
