#!/usr/bin/env python3
"""What one epoch of the device sampler costs at ml-1m shape (S-ml1m, neg_ratio 4): wall time per sample, with the
host sampler beside it, and -- under `rocprofv3 --kernel-trace --stats -- python3 tools/prof_dsampler.py` -- per kernel."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleverrec_amd import _lib, runtime  # noqa: E402
from cleverrec_amd.data import synthetic  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "pairwise"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sess = runtime.Session(0)
data = synthetic("S-ml1m")
t0 = time.perf_counter()
ms = runtime.MtSampler(sess, data)
torch.cuda.synchronize()
print("create (jump table, bitmaps): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
st = np.random.RandomState(1).get_state()
key, pos = st[1], st[2]
ts = []
for r in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cols, (key, pos) = ms.sample(kind, 4, key, pos)
    ts.append(time.perf_counter() - t0)
n = int(cols[0].numel())
print("device sampler, %s, %d instances: first %.2f ms, median %.2f ms, min %.2f ms -> %.0f M/s" % (
    kind, n, ts[0] * 1e3, np.median(ts[1:]) * 1e3, min(ts) * 1e3, n / np.median(ts[1:]) / 1e6))
print("stats", ms.stats())
ctx = _lib.Context(None)
ctx.seed(1)
t0 = time.perf_counter()
ctx.sample_pairwise(data.train_csr, data.item_nums, 4)
print("host sampler (pairwise): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
