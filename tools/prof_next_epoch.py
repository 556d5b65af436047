import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cleverrec_amd import runtime
from cleverrec_amd.data import synthetic
sess = runtime.Session(0)
data = synthetic("S-ml1m", seed=20240601)
B, NEG, D = 6144, 4, 64
g = np.random.Generator(np.random.PCG64(0))
P0 = (g.standard_normal((data.user_nums, D)) * 0.01).astype(np.float32)
Q0 = (g.standard_normal((data.item_nums, D)) * 0.01).astype(np.float32)
st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
ms = runtime.MtSampler(sess, data)
np.random.seed(1)
for e in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cu, ci, cj = ms.next_epoch("pairwise", NEG, ahead=True)
    t1 = time.perf_counter()
    st.train_epoch(cu, ci, cj, B)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print("epoch %d: next_epoch %.2f ms host, train_epoch enqueue %.2f ms, GPU done %.2f ms after that; total %.2f" % (e, (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t3-t0)*1e3))
# steps alone
torch.cuda.synchronize(); t0=time.perf_counter(); st.train_epoch(cu, ci, cj, B); torch.cuda.synchronize(); print("steps alone %.2f ms" % ((time.perf_counter()-t0)*1e3))
ms.drop(); ms.close()
