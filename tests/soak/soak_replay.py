#!/usr/bin/env python3
"""Rebuild one BPR configuration of tests/soak/soak.py from the soak's seed and the failing line's sampler seed (the soak
draws everything from one generator, so the draws of all earlier configurations are replayed, without the GPU work),
run it again on the GPU and report where the tables differ from the oracle's: which rows, how many occurrences they
had in each batch, how large the differences are.   python tests/soak/soak_replay.py --seed 11 --target 600540297"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def configs(seed):
    g = np.random.Generator(np.random.PCG64(seed))
    while True:
        U, I = int(g.integers(3, 400)), int(g.integers(6, 300))
        d = int(g.choice([4, 8, 12, 16, 32, 64, 128, 256]))
        B = int(g.choice([1, 7, 64, 333, 1000, 2048, 6144, 8192, 9000, 20000]))
        nr = int(g.integers(1, 5))
        mode = str(g.choice(["tf1", "lazy", "lazy", "sgd", "adagrad"]))  # as tests/soak/soak.py draws it
        if B > 8192 and mode == "tf1":
            mode = "lazy"
        big = int(g.random() < 0.3)
        deg_hi = max(2, min(I - nr - 1, int(g.integers(2, 40))))
        ui_train = {u: [int(x) for x in g.choice(I, size=int(g.integers(1, deg_hi + 1)), replace=False)] for u in range(U) if g.random() < 0.9}
        if not ui_train:
            continue
        scale = float(g.choice([0.01, 0.3, 1.0]))
        P0 = (g.standard_normal((U, d)) * scale).astype(np.float32)
        Q0 = (g.standard_normal((I, d)) * scale).astype(np.float32)
        s = int(g.integers(0, 2 ** 31))
        yield dict(U=U, I=I, d=d, B=B, nr=nr, mode=mode, ui_train=ui_train, scale=scale, P0=P0, Q0=Q0, seed=s, big=big)
        g.choice([1, 5, 10, 20, 24, 32])  # the eval part's draws
        g.choice(U, size=int(g.integers(1, U + 1)), replace=False)
        g.integers(0, 2)
        g.choice(["4", "8"])
        g.random()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--target", type=int, required=True)
    ap.add_argument("--big-index", type=int, default=-1, help="-1: as the soak drew it; 0 / 1: the in-LDS / the global-memory index build")
    a = ap.parse_args()
    for n, c in enumerate(configs(a.seed)):
        if c["seed"] == a.target:
            break
        if n > 200000:
            sys.exit("not found")
    big = c["big"] if a.big_index < 0 else a.big_index
    print("config #%d: U=%d I=%d d=%d B=%d nr=%d mode=%s scale=%g big_index=%d (drawn %d)" % (n, c["U"], c["I"], c["d"], c["B"], c["nr"], c["mode"], c["scale"], big, c["big"]))
    import oracle
    from cleverrec_amd import runtime
    from cleverrec_amd.data import RankingData
    sess = runtime.Session(0)
    sess.ctx.set_option("big_index", big)
    U, I, B, nr = c["U"], c["I"], c["B"], c["nr"]
    data = RankingData(U, I, c["ui_train"], {u: [0] for u in list(c["ui_train"])[:3]})
    st = runtime.BprState(sess, c["P0"], c["Q0"], lr=1e-3, reg=0.01, adam_mode=c["mode"])
    om = oracle.BprModel(c["P0"], c["Q0"], lr=1e-3, reg=0.01, adam_mode={"tf1": 0, "lazy": 1, "sgd": 2, "adagrad": 3}[c["mode"]])
    sess.ctx.seed(c["seed"])
    rng = oracle.Rng(c["seed"])
    for ep in range(2):
        u, i, j = sess.ctx.sample_pairwise(data.train_csr, I, nr)
        nb, ou, oi, oj = oracle.sample_pairwise(rng, data.ui_train, I, nr, B)
        n = min(len(u), 48 * B)
        nb = -(-n // B)
        u, i, j = u[:n], i[:n], j[:n]
        for b in range(nb):  # batch by batch, so that the first batch where a row parts ways is seen
            sl = slice(b * B, min(n, (b + 1) * B))
            before = {"gpu": (st.P.cpu().numpy().astype(np.float64), st.Q.cpu().numpy().astype(np.float64)),
                      "oracle": (om.P.astype(np.float64), om.Q.astype(np.float64))}
            loss = st.train_epoch(sess.to_dev(u[sl]), sess.to_dev(i[sl]), sess.to_dev(j[sl]), B).cpu().numpy()
            ol = oracle.train_model_py(om, 1, u[sl], i[sl], j[sl], B)[1]
            if abs(loss[0] - ol[0]) > 2e-5 * abs(ol[0]) + 1e-6:
                P64, Q64 = before["oracle"]
                x = [float(P64[a_] @ Q64[b_] - P64[a_] @ Q64[c_]) for a_, b_, c_ in zip(u[sl], i[sl], j[sl])]
                print("epoch %d batch %d: loss %.9g (gpu) vs %.9g (oracle), relative %.3g; x of the batch's triples (float64, oracle tables): %s" % (
                    ep, b, loss[0], ol[0], abs(loss[0] - ol[0]) / abs(ol[0]), np.round(x[:8], 4)))
                Pg, Qg = before["gpu"]
                xg = [float(Pg[a_] @ Qg[b_] - Pg[a_] @ Qg[c_]) for a_, b_, c_ in zip(u[sl], i[sl], j[sl])]
                print("   x from the gpu's tables:", np.round(xg[:8], 4), " max |table difference| P %.3g Q %.3g" % (np.abs(Pg - P64).max(), np.abs(Qg - Q64).max()))
                return
            for name, idx in (("P", u[sl]), ("Q", np.concatenate([i[sl], j[sl]]))):
                diff = np.abs(getattr(st, name).cpu().numpy() - getattr(om, name))
                bad = np.nonzero((diff > 1e-5 * max(c["scale"], 0.05) * 20).any(1))[0]
                if len(bad):
                    cnt = np.bincount(idx, minlength=diff.shape[0])
                    print("epoch %d batch %d %s: %d rows differ, first: row %d (occurrences in this batch %d), max diff %.3g, loss %.6g vs %.6g" % (
                        ep, b, name, len(bad), bad[0], cnt[bad[0]], diff.max(), loss[0], ol[0]))
                    g_ = getattr(st, name).cpu().numpy()[bad[0]]
                    o_ = getattr(om, name)[bad[0]]
                    np.set_printoptions(linewidth=200, precision=6)
                    k = np.nonzero(np.abs(g_ - o_) > 1e-6)[0]
                    print("   elements that differ:", k, "\n   gpu   ", g_[k], "\n   oracle", o_[k])
                    for slot in ("m" + name, "v" + name):
                        print("  ", slot, "gpu", getattr(st, slot).cpu().numpy()[bad[0]][k], "oracle", getattr(om, slot)[bad[0]][k])
                    if name == "P":  # each side's gradient of this row from its OWN tables before the step, in float64
                        for side, (P64, Q64) in before.items():
                            g64 = np.zeros(P64.shape[1])
                            for q in np.nonzero(u[sl] == bad[0])[0]:
                                p_, qi, qj = P64[bad[0]], Q64[i[sl][q]], Q64[j[sl][q]]
                                x = p_ @ qi - p_ @ qj
                                g64 += -(1.0 / (1.0 + np.exp(x))) * (qi - qj) + 0.01 * p_
                            print("   %-6s float64 gradient of those elements from its own tables:" % side, g64[k], "-> m = 0.1 g =", 0.1 * g64[k])
                    t = np.nonzero(idx[:len(u[sl])] == bad[0])[0] if name == "P" else []
                    print("   triples of this row in the batch (u, i, j):", [(int(u[sl][q]), int(i[sl][q]), int(j[sl][q])) for q in t])
                    return
    print("no row differs by more than the tight bound when the epochs are run batch by batch")


if __name__ == "__main__":
    main()
