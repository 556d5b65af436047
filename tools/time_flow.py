"""Times the tf1 BPR step at the headline shape (S-ml1m, d = 64, B = 6144): bpr_steps_flow against bpr_step_fused, as calls of
20 batches (the driver's timed region) and of a whole epoch.  python tools/time_flow.py [--d 64] [--reps 200]"""
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
ap.add_argument("--d", type=int, default=64)
ap.add_argument("--reps", type=int, default=100)
ap.add_argument("--steps", type=int, nargs="*", default=[20, 64, 455])
ap.add_argument("--flow", type=int, nargs="*", default=[-1, 0])
ap.add_argument("--abl", type=int, nargs="*", default=[0])
a = ap.parse_args()
B = 6144
sess = runtime.Session(0)
data = synthetic("S-ml1m", seed=20240601)
g = torch.Generator().manual_seed(0)
P0 = torch.empty(data.user_nums, a.d).normal_(0, 0.01, generator=g).numpy()
Q0 = torch.empty(data.item_nums, a.d).normal_(0, 0.01, generator=g).numpy()
sess.ctx.seed(1)
u, i, j = (sess.to_dev(x) for x in sess.ctx.sample_pairwise(data.train_csr, data.item_nums, 4))
for flow, abl in [(f_, a_) for f_ in a.flow for a_ in (a.abl if f_ != 0 else [0])]:
    sess.ctx.set_option("flow", flow)
    sess.ctx.set_option("no_call_tail", int(os.environ.get("CREC_NO_CALL_TAIL", "0")))  # tools/time_call_tail.sh
    sess.ctx.set_option("flow_abl", abl)
    st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
    for steps in a.steps:
        n = steps * B
        out = sess.empty((steps,), torch.float32)
        for _ in range(5):
            st.train_epoch(u[:n], i[:n], j[:n], B, out=out)
        torch.cuda.synchronize()
        reps = max(3, a.reps * 20 // steps)
        t0 = time.perf_counter()
        for _ in range(reps):
            st.train_epoch(u[:n], i[:n], j[:n], B, out=out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        # one call at a time (what a timed region of exactly `steps` steps sees)
        ts = []
        for _ in range(min(reps, 50)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            st.train_epoch(u[:n], i[:n], j[:n], B, out=out)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        sess.ctx.profile(True)
        st.train_epoch(u[:n], i[:n], j[:n], B, out=out)
        st.train_epoch(u[:n], i[:n], j[:n], B, out=out)
        prof = sess.ctx.profile_read()
        units = sess.ctx.profile_units("bpr_steps_flow")
        sess.ctx.profile(False)
        kern = {k: (round(ms / n_ * 1e3, 2), n_) for k, (ms, n_) in prof.items() if n_}
        print("abl=%d flow=%2d steps=%4d  back-to-back %.2f us/step (%.0f M triples/s)  single call median %.2f us/step  kernels(us, launches)=%s flow_us_per_step=%s"
              % (abl, flow, steps, dt / steps * 1e6, steps * B / dt / 1e6, float(np.median(ts)) / steps * 1e6, kern,
                 round(prof["bpr_steps_flow"][0] / units * 1e3, 3) if units else None), flush=True)
