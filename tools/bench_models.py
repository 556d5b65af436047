#!/usr/bin/env python3
"""Throughput of the other graphs of the path at their BASELINE.json configs (single MI355X):
  C3  NeuMF on S-ml1m, d=64, layers [128,64,32], batch 6144, neg_ratio 4   -> instances/s
  C4  FM, F=2 one-hot fields (user, U+item), d=32, N = U+I features, batch 6144 -> rows/s
  a11 CML on S-ml1m, d=128, neg_ratio 20, batch 6144                          -> pairs/s
  f4  LRML on S-ml1m, d=128, mem_size 50, neg_ratio 1, batch 6144            -> triples/s, distance matrix users/s
  f4  TransCF on S-ml1m, d=64, neg_ratio 4, batch 6144 (100 batches)          -> triples/s, distance matrix users/s
Index streams come from the bit-exact host samplers; everything is resident in HBM when timed."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from cleverrec_amd import runtime
    from cleverrec_amd.data import synthetic
    sess = runtime.Session(0)
    data = synthetic("S-ml1m")
    U, I, B = data.user_nums, data.item_nums, 6144
    g = np.random.Generator(np.random.PCG64(0))
    rnd = lambda *s: (g.standard_normal(s) * 0.05).astype(np.float32)
    out = {}

    def timed(fn, n_units, reps=2):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        return {"per_s": n_units / dt, "ms_per_epoch": dt * 1e3, "us_per_step": dt / -(-n_units // B) * 1e6}

    # ---- NeuMF (C3) and GMF
    sess.ctx.seed(1)
    u, i, y = sess.ctx.sample_pointwise(data.train_csr, I, 4)
    du, di, dy = sess.to_dev(u), sess.to_dev(i), sess.to_dev(y)
    Ws, bs, w = [], [], 128
    for _ in range(3):
        Ws.append(rnd(w, w // 2))
        bs.append(rnd(w // 2))
        w //= 2
    st = runtime.NcfState(sess, "NeuMF", {"P_gmf": rnd(U, 64), "Q_gmf": rnd(I, 64), "P_mlp": rnd(U, 64), "Q_mlp": rnd(I, 64)},
                          Ws, bs, rnd(64 + 16), lr=1e-3, reg_gmf=1e-2, reg_mlp=1e-3)
    out["NeuMF_C3"] = dict(timed(lambda: st.train_epoch(du, di, dy, B), len(u)), unit="instances", instances=len(u))
    st = runtime.NcfState(sess, "GMF", {"P_gmf": rnd(U, 64), "Q_gmf": rnd(I, 64)}, [], [], rnd(64), lr=1e-3, reg_gmf=1e-2)
    out["GMF"] = dict(timed(lambda: st.train_epoch(du, di, dy, B), len(u)), unit="instances")
    users = sess.to_dev(data.test_users)
    st = runtime.NcfState(sess, "NeuMF", {"P_gmf": rnd(U, 64), "Q_gmf": rnd(I, 64), "P_mlp": rnd(U, 64), "Q_mlp": rnd(I, 64)},
                          Ws, bs, rnd(64 + 16), lr=1e-3, reg_gmf=1e-2, reg_mlp=1e-3)
    sp, si = sess.to_dev(data.seen_csr[0]), sess.to_dev(data.seen_csr[1])

    def neumf_eval():
        for t in range(0, len(users), 1024):
            cur = users[t:t + 1024]
            sc = runtime.ncf_scores(sess, st, cur)
            runtime.topk_scores(sess, sc, cur, sp, si, 20)
    r = timed(neumf_eval, len(users))
    out["NeuMF_eval"] = {"users_per_s": r["per_s"], "ms": r["ms_per_epoch"]}

    # ---- FM (C4): F = 2 (user, U + item), labels 1..5
    n = 800_000
    xu = g.integers(0, U, n)
    xi = g.integers(0, I, n)
    x_idx = np.stack([xu, U + xi], 1).astype(np.int32)
    x_val = np.ones((n, 2), np.float32)
    yy = g.integers(1, 6, n).astype(np.float32)
    fm = runtime.FmState(sess, np.zeros(1, np.float32), rnd(U + I + 1), rnd(U + I + 1, 32), lr=1e-3, reg=1e-3)
    dxi, dxv, dyy = sess.to_dev(x_idx), sess.to_dev(x_val), sess.to_dev(yy)
    out["FM_C4"] = dict(timed(lambda: fm.train_epoch(dxi, dxv, dyy, B), n), unit="rows")

    # ---- CML: d = 128, 20 negatives
    sess.ctx.seed(1)
    u, i, neg = sess.ctx.sample_cml(data.train_csr, I, 20)
    cml = runtime.CmlState(sess, rnd(U, 128), rnd(I, 128), lr=1e-3, reg=10.0, margin=1.0)
    du, di, dn = sess.to_dev(u), sess.to_dev(i), sess.to_dev(neg)
    out["CML"] = dict(timed(lambda: cml.train_epoch(du, di, dn, B), len(u)), unit="pairs")

    # ---- LRML (f4): the shipped conf -- d = 128, 50 memory slots, one negative per positive, hinge
    sess.ctx.seed(1)
    u, i, j = sess.ctx.sample_pairwise(data.train_csr, I, 1)
    lr_ = runtime.LrmlState(sess, rnd(U, 128), rnd(I, 128), rnd(128, 50), rnd(50, 128), lr=1e-3, reg=1e-3, margin=0.2)
    du, di, dj = sess.to_dev(u), sess.to_dev(i), sess.to_dev(j)
    out["LRML"] = dict(timed(lambda: lr_.train_epoch(du, di, dj, B), len(u)), unit="triples")
    users = sess.to_dev(np.arange(1024, dtype=np.int32))
    lr_.scores(users)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        lr_.scores(users)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    out["LRML_scores"] = {"users_per_s": 1024 / ms * 1e3, "ms_per_1024_users": ms}

    # ---- TransCF (f4): the shipped conf -- d = 64, four negatives per positive, hinge + both regularisers
    from cleverrec_amd.utils.tools import get_sp_mat
    sess.ctx.seed(1)
    u, i, j = sess.ctx.sample_pairwise(data.train_csr, I, 4)
    tc = runtime.TranscfState(sess, rnd(U, 64), rnd(I, 64), get_sp_mat(data), lr=1e-3, reg1=0.1, reg2=0.01, margin=0.5)
    n = 100 * B
    du, di, dj = sess.to_dev(u[:n]), sess.to_dev(i[:n]), sess.to_dev(j[:n])
    out["TransCF"] = dict(timed(lambda: tc.train_epoch(du, di, dj, B), n), unit="triples")
    tc.scores(users)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        tc.scores(users)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    out["TransCF_scores"] = {"users_per_s": 1024 / ms * 1e3, "ms_per_1024_users": ms}

    # ---- SBPR / CUNE_BPR (f4): the shipped confs -- d = 128, B = 6144; social / latent-friend items drawn like positives
    n = 100 * B
    us, is_, ks, js = (sess.to_dev(g.integers(0, m, n).astype(np.int32)) for m in (U, I, I, I))
    suk = sess.to_dev(g.integers(1, 4, n).astype(np.float32))
    sb = runtime.SbprState(sess, rnd(U, 128), rnd(I, 128), np.zeros(I + 1, np.float32), lr=1e-3, reg=0.05)
    out["SBPR"] = dict(timed(lambda: sb.train_epoch(us, is_, ks, js, suk, B), n), unit="quadruples")
    cu = runtime.CuneState(sess, rnd(U, 128), rnd(I, 128), np.zeros(I + 1, np.float32), 0.0, lr=1e-3, reg=0.01)
    out["CUNE_BPR"] = dict(timed(lambda: cu.train_epoch(us, is_, ks, js, B), n), unit="quadruples")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
