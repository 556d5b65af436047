"""Per-wave timeline of CML's update launch (apply_all<32>: one wave per table row, the user table's 6 040 rows then the item table's
3 706) from a -DCREC_ROWS_STAMPS build (tools/rows_stamps.sh): when the waves begin and end, what they wait for, how that depends on
the row's number of slots.  Times in microseconds from the first wave's begin (100 MHz clock: 10 ns steps)."""
import ctypes as C
import os
import sys
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
    cml = runtime.CmlState(sess, rnd(U, 128), rnd(I, 128), lr=1e-3, reg=10.0, margin=1.0)
    du, di, dn = sess.to_dev(u), sess.to_dev(i), sess.to_dev(neg)
    cml.train_epoch(du, di, dn, B)
    cml.train_epoch(du[:10 * B], di[:10 * B], dn[:10 * B], B)  # the stamps are those of the last launch: a full batch behind nine others
    torch.cuda.synchronize()
    n = 8 * 16384
    raw = np.zeros(n, np.uint64)
    rc = sess.lib.crec_debug_rows_stamps(raw.ctypes.data_as(C.c_void_p), C.c_size_t(n))
    assert rc == 0, rc
    st = raw.reshape(-1, 8).astype(np.int64)
    t0 = st[:U + I + 4, 0][st[:U + I + 4, 0] > 0].min()
    us = lambda x: (x - t0) / 100.0
    # (the item table's blocks lead the grid, four rows per block: the user table's waves start at the next multiple of four)
    i_end = -(-I // 4) * 4
    for name, lo, hi in (("item table (3706 rows, per-step index, 6 chunks)", 0, I), ("user table (6040 rows, per-epoch index, 1 chunk)", i_end, i_end + U)):
        s = st[lo:hi]
        beg, end, slots = us(s[:, 0]), us(s[:, 1]), s[:, 2]
        dur = end - beg
        q = lambda x: "min %.2f  median %.2f  p90 %.2f  max %.2f" % (x.min(), np.median(x), np.percentile(x, 90), x.max())
        print(name)
        print("  begin   ", q(beg))
        print("  end     ", q(end))
        print("  duration", q(dur))
        ph = [("first segment arrived", us(s[:, 4]) - beg), ("walk done", us(s[:, 5]) - us(s[:, 4])), ("row arrived", us(s[:, 6]) - us(s[:, 5])),
              ("m, v arrived + optimizer", us(s[:, 7]) - us(s[:, 6])), ("stores retired", end - us(s[:, 7]))]
        for nm, x in ph:
            print("    %-26s" % nm, q(x))
        print("  duration by slots of the row:")
        for a, b in ((0, 0), (1, 1), (2, 3), (4, 7), (8, 15), (16, 31), (32, 10 ** 6)):
            m = (slots >= a) & (slots <= b)
            if m.any():
                print("    %3d..%-6s rows %5d  walk median %.2f max %.2f   duration median %.2f max %.2f   begin median %.2f" % (
                    a, b if b < 10 ** 6 else "", int(m.sum()), np.median(ph[1][1][m]), ph[1][1][m].max(), np.median(dur[m]), dur[m].max(), np.median(beg[m])))


main()
