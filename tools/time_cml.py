"""CML step alone (S-ml1m, d = 128, 20 negatives, batch 6144): microseconds per step and per kernel (ctx profile).
Env: CML_USERS=n folds the user ids onto n table rows."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


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
    sess.ctx.seed(1)
    u, i, neg = sess.ctx.sample_cml(data.train_csr, I, 20)
    if os.environ.get("CML_USERS"):  # fewer table rows (user ids folded): how the update launch's time depends on its number of row waves
        U = int(os.environ["CML_USERS"])
        u = (u % U).astype(u.dtype)
    cml = runtime.CmlState(sess, rnd(U, 128), rnd(I, 128), lr=1e-3, reg=10.0, margin=1.0)
    du, di, dn = sess.to_dev(u), sess.to_dev(i), sess.to_dev(neg)
    fn = lambda: cml.train_epoch(du, di, dn, B)
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    steps = -(-len(u) // B)
    sess.ctx.profile(True)
    fn()
    torch.cuda.synchronize()
    prof = {k: round(ms / n * 1e3, 2) for k, (ms, n) in sess.ctx.profile_read().items() if n}
    sess.ctx.profile(False)
    # how many slots the batch's heaviest item rows hold: positives + the chosen (nearest) negatives, from the tables as they are now
    P, Q = cml.P.cpu().numpy(), cml.Q.cpu().numpy()
    ub, ib, nb_ = u[:B].astype(np.int64), i[:B].astype(np.int64), neg[:B].astype(np.int64)
    dist = ((P[ub][:, None, :] - Q[nb_]) ** 2).sum(-1)
    kmin = nb_[np.arange(B), dist.argmin(1)]
    cnt = np.bincount(np.concatenate([ib, kmin]), minlength=I)
    heavy = {"max_slots_per_item_row": int(cnt.max()), "rows_over_12": int((cnt > 12).sum()), "rows_over_64": int((cnt > 64).sum()),
             "max_chosen_negative": int(np.bincount(kmin, minlength=I).max())}
    print(json.dumps({"item_rows": heavy, "users": U, "us_per_step": round(dt / steps * 1e6, 2), "kernels_us": prof}))


main()
