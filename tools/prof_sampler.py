import sys, time
sys.path.insert(0, '/root/repo')
import torch
from cleverrec_amd import _lib
from cleverrec_amd.data import synthetic
data = synthetic("S-ml1m")
ctx = _lib.Context(-1)
for k in range(3):
    ctx.seed(1); t0 = time.perf_counter(); u, i, j = ctx.sample_pairwise(data.train_csr, data.item_nums, 4); print("total %.1f ms" % ((time.perf_counter() - t0) * 1e3), len(u))
