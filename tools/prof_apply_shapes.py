"""The update launch (apply_all) and the index launches of the SBPR step by HIP events at a few table shapes and batch sizes:
what is sweep (B = 64: nearly no slots) and what is the slot walk (B = 6144).  python tools/prof_apply_shapes.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleverrec_amd import runtime
sess = runtime.Session(0)
g = np.random.Generator(np.random.PCG64(0))
rnd = lambda *s: (g.standard_normal(s) * 0.05).astype(np.float32)
for U, I, d, B in ((6040, 3706, 128, 6144), (6040, 3706, 128, 64), (6040, 3706, 64, 6144), (6040, 3706, 64, 64), (60400, 37060, 128, 6144), (604, 370, 128, 64)):
    n = 100 * B
    us, is_, ks, js = (sess.to_dev(g.integers(0, m, n).astype(np.int32)) for m in (U, I, I, I))
    suk = sess.to_dev(g.integers(1, 4, n).astype(np.float32))
    sb = runtime.SbprState(sess, rnd(U, d), rnd(I, d), np.zeros(I + 1, np.float32), lr=1e-3, reg=0.05)
    for _ in range(2):
        sb.train_epoch(us, is_, ks, js, suk, B)
    torch.cuda.synchronize()
    sess.ctx.profile(True)
    sb.train_epoch(us, is_, ks, js, suk, B)
    torch.cuda.synchronize()
    p = sess.ctx.profile_read()
    sess.ctx.profile(False)
    print(U, I, d, B, {k: (round(v[0] / max(v[1], 1) * 1e3, 2), v[1]) for k, v in p.items() if v[1]})
