#!/usr/bin/env python3
"""Per-wave timeline of one bpr_step_fused launch (debug build of bpr_train.hip with -DCREC_STAMPS, see
tools/wave_stamps.sh): when each wave started and ended (100 MHz wall clock), its length in shader clocks, its XCD.
Prints the launch's span, the distribution of start offsets and durations, and the slowest waves by class."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from cleverrec_amd import _lib, runtime
    from cleverrec_amd.data import synthetic
    B, D = 6144, 64
    sess = runtime.Session(0)
    data = synthetic("S-ml1m", seed=20240601)
    P0 = torch.empty(data.user_nums, D).normal_(0, 0.01, generator=torch.Generator().manual_seed(1000)).numpy()
    Q0 = torch.empty(data.item_nums, D).normal_(0, 0.01, generator=torch.Generator().manual_seed(0)).numpy()
    st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
    sess.ctx.seed(1)
    u, i, j = sess.ctx.sample_pairwise(data.train_csr, data.item_nums, 4)
    steps = int(os.environ.get("STEPS", "40"))
    du, di, dj = (sess.to_dev(x[:steps * B]) for x in (u, i, j))
    lib0 = ctypes.CDLL(_lib.LIB_PATH)
    st.train_epoch(du, di, dj, B)
    torch.cuda.synchronize()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    n = 8 * 16384
    buf = (ctypes.c_ulonglong * n)()
    rc = lib.crec_debug_stamps(buf, ctypes.c_size_t(n))
    assert rc == 0, rc
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
    marks_idx = a[16000:16014, 0].reshape(7, 2)  # bpr_index_batches, workgroups 0 (user side) and 1 (item side)
    if marks_idx.min() > 0:
        names = ("fill + sort", "run starts", "heavy rows", "wide groups", "per-rank outputs, row segments", "occurrence records")
        for side, sname in enumerate(("user", "item")):
            d = np.diff(marks_idx[:, side]) * 10.0 / 1e3
            print("index kernel, %s-side workgroup: %s = %.1f us" % (sname, ", ".join("%s %.1f" % (nm, x) for nm, x in zip(names, d)), d.sum()))
    a[16000:] = 0
    n_wide = int(((a[:, 7] & 4) != 0).sum())  # waves of the wide groups' workgroups come first
    rowflag = ((a[:, 7] & 2) != 0) & ((a[:, 7] & 4) == 0)
    first_row = n_wide + int(np.argmax(rowflag[n_wide:]))
    live = (a[:, 1] > 0) & (rowflag | (np.arange(len(a)) < first_row))  # drop the grid's padding waves
    a = a[live]
    ids = np.nonzero(live)[0]
    t0 = a[:, 0].min()
    start = (a[:, 0] - t0) * 10.0  # ns
    end = (a[:, 1] - t0) * 10.0
    dur = end - start
    print("waves %d, span %.2f us (first start to last end)" % (len(a), end.max() / 1e3))
    q = [0, 10, 50, 90, 99, 100]
    print("start offset ns  pct", q, np.percentile(start, q).round(0))
    print("duration ns      pct", q, np.percentile(dur, q).round(0))
    print("end ns           pct", q, np.percentile(end, q).round(0))
    print("shader clocks per wave: mean %.0f, max %d" % (a[:, 2].mean(), a[:, 2].max()))
    WU = (data.user_nums + 3) // 4
    heavy = first_row  # wave ids below this: wide groups' waves, then chunk waves
    WI = (data.item_nums + 3) // 4
    items_first = os.environ.get("ITEMS_FIRST") == "1"
    if items_first:
        cls = np.where(ids < heavy, 0, np.where(ids < heavy + WI, 2, 1))
    else:
        cls = np.where(ids < heavy, 0, np.where(ids < heavy + WU, 1, 2))
    cls = np.where(ids < n_wide, 3, cls)
    for c, name in enumerate(("chunk", "user", "item", "wide")):
        m = cls == c
        if m.any():
            print("%-5s waves %5d: start p50 %6.0f  dur p50 %6.0f p99 %6.0f max %6.0f  end max %6.0f" % (
                name, m.sum(), np.median(start[m]), np.median(dur[m]), np.percentile(dur[m], 99), dur[m].max(), end[m].max()))
    # occurrence records each row wave walked in this (the last) batch
    lb = slice((steps - 1) * B, steps * B)
    cu = np.bincount(u[lb], minlength=data.user_nums)
    ci = np.bincount(i[lb], minlength=data.item_nums) + np.bincount(j[lb], minlength=data.item_nums)
    def per_wave(c):
        c = np.concatenate([c, np.zeros((-len(c)) % 4, dtype=c.dtype)])
        return c.reshape(-1, 4).sum(1), c.reshape(-1, 4).max(1)
    (su, mu), (si, mi) = per_wave(cu), per_wave(ci)
    recs = np.zeros(len(a), dtype=np.int64)
    rmax = np.zeros(len(a), dtype=np.int64)
    mu_ = cls == 1
    uoff = heavy + (WI if items_first else 0)
    recs[mu_] = su[ids[mu_] - uoff]; rmax[mu_] = mu[ids[mu_] - uoff]
    mi_ = cls == 2
    ioff = heavy + (0 if items_first else WU)
    recs[mi_] = si[np.minimum(ids[mi_] - ioff, len(si) - 1)]; rmax[mi_] = mi[np.minimum(ids[mi_] - ioff, len(si) - 1)]
    print("duration by records walked (row waves): records, waves, median dur, max dur")
    for lo, hi in ((0, 0), (1, 4), (5, 8), (9, 16), (17, 24), (25, 32), (33, 48), (49, 64), (65, 96), (97, 1 << 30)):
        m = (cls > 0) & (recs >= lo) & (recs <= hi)
        if m.any():
            print("  %3d..%-4s %5d  %6.0f %6.0f" % (lo, hi if hi < 1 << 30 else "", m.sum(), np.median(dur[m]), dur[m].max()))
    span_heavy = a[:, 7].copy()
    a[:, 7] = 0
    marks = np.where(a[:, 4:8] > 0, (a[:, 4:8] - t0) * 10.0, np.nan)  # segment, records, gradient, m/v  (ns from launch start)
    for c, name in ((1, "user"), (2, "item")):
        m = cls == c
        print("%s waves, median ns since launch start: start %4.0f | segment %5.0f | records %5.0f | gradient %5.0f | m,v %5.0f | end %5.0f" % (
            (name, np.median(start[m])) + tuple(np.nanmedian(marks[m], 0)) + (np.median(end[m]),)))
    walked = span_heavy >> 32
    for c, name in ((1, "user"), (2, "item")):
        m = cls == c
        print("%s waves: records by the host count %d, walked by the waves %d, waves where the two differ %d" % (
            name, recs[m].sum(), walked[m].sum(), (recs[m] != walked[m]).sum()))
    m = np.nonzero(cls == 1)[0][:24]
    print("first user waves: host count / walked:", " ".join("%d/%d" % (recs[k], walked[k]) for k in m))
    m = np.nonzero(cls == 2)[0][:24]
    print("first item waves: host count / walked:", " ".join("%d/%d" % (recs[k], walked[k]) for k in m))
    order = np.argsort(-end)[:12]
    print("last to end: wave id, class, xcd, start, [segment, records, gradient, m/v], end (ns), records, max per row")
    for k in order:
        print("  %5d %d %d %6.0f %s %6.0f %4d %4d | walked range %d, heavy slot %d" % (ids[k], cls[k], a[k, 3], start[k], np.round(marks[k][:3]), end[k], recs[k], rmax[k], span_heavy[k] >> 32, span_heavy[k] & 1))
    ub, ib, jb = u[lb], i[lb], j[lb]
    for k in order[:3]:
        if cls[k] != 2:
            continue
        r0 = (ids[k] - ioff) * 4
        print("records of item wave %d (rows %d..%d): (u, i, j, batch count of u, of i, of j)" % (ids[k], r0, r0 + 3))
        sel = np.nonzero(((ib >= r0) & (ib < r0 + 4)) | ((jb >= r0) & (jb < r0 + 4)))[0]
        for t in sel:
            print("   ", ub[t], ib[t], jb[t], cu[ub[t]], ci[ib[t]], ci[jb[t]])
    for x in range(8):
        m = a[:, 3] == x
        if m.any():
            print("xcd %d: waves %4d first start %5.0f last end %6.0f" % (x, m.sum(), start[m].min(), end[m].max()))
    # start offset by launch order
    blocks = ids // 4
    for lo in range(0, blocks.max() + 1, 64):
        m = (blocks >= lo) & (blocks < lo + 64)
        print("workgroups %4d..%4d: start min %5.0f median %5.0f  end max %6.0f" % (lo, lo + 63, start[m].min(), np.median(start[m]), end[m].max()))


if __name__ == "__main__":
    main()
