"""bpr_steps_flow beside other work on the GPU: epochs of next_epoch() (the next sample drawn on a side stream) + train_epoch,
and epochs beside a stream of unrelated kernels that fill the chip.  Every wave of a flow launch must get onto the chip while
others spin for it; prints the seconds per epoch and any give-up.  python tools/flow_stress.py [--epochs 12]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cleverrec_amd import runtime  # noqa: E402
from cleverrec_amd.data import synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--epochs", type=int, default=12)
ap.add_argument("--mode", default="sampler,side,plain")
a = ap.parse_args()
B, D, NEG = 6144, 64, 4
sess = runtime.Session(0)
data = synthetic("S-ml1m", seed=20240601)
g = torch.Generator().manual_seed(0)
P0 = torch.empty(data.user_nums, D).normal_(0, 0.01, generator=g).numpy()
Q0 = torch.empty(data.item_nums, D).normal_(0, 0.01, generator=g).numpy()
for mode in a.mode.split(","):
    st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
    ts, err = [], None
    try:
        if mode == "sampler":
            ms = runtime.MtSampler(sess, data)
            np.random.seed(1)
            for e in range(a.epochs):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                cu, ci, cj = ms.next_epoch("pairwise", NEG, ahead=True)
                st.train_epoch(cu, ci, cj, B)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            ms.drop()
            ms.close()
        else:
            sess.ctx.seed(1)
            u, i, j = (sess.to_dev(x) for x in sess.ctx.sample_pairwise(data.train_csr, data.item_nums, NEG))
            side = torch.cuda.Stream()
            x = torch.randn(8192, 8192, device="cuda")
            for e in range(a.epochs):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if mode == "side":
                    with torch.cuda.stream(side):
                        for _ in range(6):
                            y = x @ x  # ~1 ms each: workgroups of another kernel on every CU while the flow launches start
                st.train_epoch(u, i, j, B)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
    except Exception as e_:  # noqa: BLE001
        err = repr(e_)
    print("%-8s epochs %s ms %s" % (mode, " ".join("%.1f" % (t * 1e3) for t in ts), ("ERROR " + err) if err else "ok"), flush=True)
