import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from cleverrec_amd import runtime
import test_gpu_window as T
sess = runtime.Session(0)
U, I, d, B, nb, zipf, mode = T.CASES[int(os.environ.get("CASE", "0"))]
W = int(os.environ.get("W", "2"))
u, i, j = T._stream(U + d + B, U, I, nb * B + B // 3 + 1, zipf)
nbat = int(os.environ.get("NB", "4"))
n = min(nbat * B, len(u))
ref = T._run(sess, 0, mode, U, I, d, B, u[:n], i[:n], j[:n])
print("ref max |Q| %g |P| %g" % (np.abs(ref[2]).max(), np.abs(ref[1]).max()))
for rep in range(6):
    got = T._run(sess, W, mode, U, I, d, B, u[:n], i[:n], j[:n])
    msg = []
    for x, y, name in zip(ref, got, ("loss", "P", "Q", "mP", "mQ", "vP", "vQ")):
        if not np.array_equal(x, y):
            if x.ndim == 2:
                rows = np.nonzero((x != y).any(1))[0]
                info = []
                for r in rows[:4]:
                    arr_i, arr_j, arr_u = i[:n], j[:n], u[:n]
                    occ = [(b, int(((arr_u if name in ("P", "mP", "vP") else arr_i)[b * B:(b + 1) * B] == r).sum()), int((arr_j[b * B:(b + 1) * B] == r).sum()) if name[-1] == "Q" else 0) for b in range(nbat)]
                    info.append("row %d occ(batch, i|u, j) %s got max %g" % (r, [o for o in occ if o[1] or o[2]], np.abs(y[r]).max()))
                msg.append("%s: %d rows differ: %s" % (name, len(rows), "; ".join(info)))
            else:
                msg.append("%s differs at %s" % (name, np.nonzero(x != y)[0][:8]))
    print("rep", rep, "OK" if not msg else "\n   ".join(msg), flush=True)
