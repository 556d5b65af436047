#!/usr/bin/env python3
"""tf1 BPR step at a few other shapes (table sizes, widths): us/step over whole epochs and the kernels' own durations."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
from cleverrec_amd import runtime
from cleverrec_amd.data import synthetic
for name, d in (("S-ml100k", 32), ("S-ml100k", 64), ("S-ml1m", 128), ("S-ml1m", 16)):
    sess = runtime.Session(0)
    data = synthetic(name)
    g = torch.Generator().manual_seed(0)
    P0 = torch.empty(data.user_nums, d).normal_(0, 0.01, generator=g).numpy()
    Q0 = torch.empty(data.item_nums, d).normal_(0, 0.01, generator=g).numpy()
    st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
    sess.ctx.seed(1)
    u, i, j = (sess.to_dev(a) for a in sess.ctx.sample_pairwise(data.train_csr, data.item_nums, 4))
    B = 6144
    nb = len(u) // B
    for _ in range(5):
        st.train_epoch(u[:nb * B], i[:nb * B], j[:nb * B], B)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = max(1, 20000 // nb)
    for _ in range(reps):
        st.train_epoch(u[:nb * B], i[:nb * B], j[:nb * B], B)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sess.ctx.profile(True)
    st.train_epoch(u[:nb * B], i[:nb * B], j[:nb * B], B)
    torch.cuda.synchronize()
    prof = {k: round(ms / n * 1e3, 2) for k, (ms, n) in sess.ctx.profile_read().items() if n}
    sess.ctx.profile(False)
    print("   per launch (us):", prof)
    print("%s d=%d: %d x %d, %d batches/epoch: %.2f us/step, %.0f M triples/s" % (name, d, data.user_nums, data.item_nums, nb, dt / (reps * nb) * 1e6, reps * nb * B / dt / 1e6))
