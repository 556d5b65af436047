"""lazy_window under random shapes: tables, losses and slots after W batches per launch against a launch per batch, bit for bit
(tests/test_gpu_window.py holds the fixed cases); and beside a stream of unrelated kernels that fill the chip (a window launch's
waves wait only for workgroups dispatched before them: it must neither dead-lock nor give up there).
python tools/window_stress.py [--seconds 120] [--seed 0]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cleverrec_amd import runtime  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
sess = runtime.Session(0)
g = np.random.Generator(np.random.PCG64(a.seed))
side = torch.cuda.Stream()
x = torch.randn(4096, 4096, device="cuda")


def run(W, mode, P0, Q0, u, i, j, B, calls, busy):
    sess.ctx.set_option("lazy_window", W)
    try:
        st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode=mode)
        nb = -(-len(u) // B)
        cuts = [round(c * nb / calls) * B for c in range(calls)] + [len(u)]
        du, di, dj = (sess.to_dev(v) for v in (u, i, j))
        if busy:
            with torch.cuda.stream(side):
                for _ in range(4):
                    y = x @ x  # noqa: F841
        losses = [st.train_epoch(du[p:q], di[p:q], dj[p:q], B).cpu().numpy() for p, q in zip(cuts[:-1], cuts[1:]) if q > p]
        st.flush()
        torch.cuda.synchronize()
        return [np.concatenate(losses)] + [getattr(st, n).cpu().numpy() for n in ("P", "Q", "mP", "mQ", "vP", "vQ")]
    finally:
        sess.ctx.set_option("lazy_window", -1)


t_end, n, bad = time.time() + a.seconds, 0, 0
while time.time() < t_end:
    U = int(g.choice([30, 300, 5000, 60000, 400000]))
    I = int(g.choice([1, 7, 200, 3000, 50000, 300000]))
    d = int(g.choice([4, 8, 16, 20, 32, 64, 100, 128, 256]))
    B = int(g.choice([64, 256, 1000, 2048, 4096, 6144, 8192]))
    nb = int(g.integers(2, 40))
    W = int(g.choice([2, 3, 4, 8, 12, 16]))
    mode = str(g.choice(["lazy", "lazy", "sgd", "adagrad", "deferred"]))
    zipf = float(g.choice([1.05, 1.2, 1.5, 0.0]))
    tail = int(g.integers(0, B))
    nn = nb * B + tail
    u = g.integers(0, U, nn).astype(np.int32)
    i = (np.minimum(g.zipf(zipf, nn) - 1, I - 1) if zipf else g.integers(0, I, nn)).astype(np.int32)
    j = g.integers(0, I, nn).astype(np.int32)
    P0, Q0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32), (g.standard_normal((I, d)) * 0.01).astype(np.float32)
    calls = int(g.integers(1, 4))
    busy = bool(g.integers(0, 2))
    cfg = dict(U=U, I=I, d=d, B=B, nb=nb, W=W, mode=mode, zipf=zipf, tail=tail, calls=calls, busy=busy)
    try:
        ref = run(0, mode, P0, Q0, u, i, j, B, 1, False)
        got = run(W, mode, P0, Q0, u, i, j, B, calls, busy)
        same = all(np.array_equal(p, q) for p, q in zip(ref, got))
    except Exception as e:  # noqa: BLE001
        same = False
        cfg["error"] = repr(e)
    n += 1
    if not same:
        bad += 1
        print("MISMATCH", cfg, flush=True)
    if n % 25 == 0:
        print("%d configs, %d mismatches" % (n, bad), flush=True)
print("window_stress: %d configs, %d mismatches (seed %d)" % (n, bad, a.seed), flush=True)
sys.exit(1 if bad else 0)
