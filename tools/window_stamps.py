#!/usr/bin/env python3
"""Timeline of ONE lazy_window launch at S-big shape (debug build with -DCREC_STAMPS: STAMP_TOOL=tools/window_stamps.py
bash tools/wave_stamps.sh): per batch of the launch, when its first wave started, when its waves' waits ended, when the hottest
item row's (segment start 0) contributors drew their tickets, when its finisher started / had its sum / left its stamp, and when
the batch's last stamp and last wave ended.  Times in us from the launch's first wave."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from cleverrec_amd import _lib, runtime
    W = int(os.environ.get("W", "8"))
    users, items, dim, batch = int(os.environ.get("USERS", "10000000")), 1_000_000, 128, 6144
    sess = runtime.Session(0)
    sess.ctx.set_option("lazy_window", W)
    if os.environ.get("ABL"):
        sess.ctx.set_option("flow_abl", int(os.environ["ABL"]))
    dev = sess.device
    g = torch.Generator(device=dev).manual_seed(0)
    st = runtime.BprState.__new__(runtime.BprState)
    st.sess = sess
    st.P = torch.empty((users, dim), dtype=torch.float32, device=dev).normal_(0, 0.01, generator=g)
    st.Q = torch.empty((items, dim), dtype=torch.float32, device=dev).normal_(0, 0.01, generator=g)
    st.mP, st.vP, st.mQ, st.vQ = torch.zeros_like(st.P), torch.zeros_like(st.P), torch.zeros_like(st.Q), torch.zeros_like(st.Q)
    st.tables = _lib.BprTables(*(t.data_ptr() for t in (st.P, st.Q, st.mP, st.vP, st.mQ, st.vQ)), users, items, dim)
    st.adam = _lib.Adam(1e-3, 0.9, 0.999, 1e-8, 0.01, _lib.ADAM_LAZY, 0.9, 0.999, 0)
    n = (64 + W) * batch
    u = torch.randint(0, users, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    w = 1.0 / torch.arange(1, items + 1, dtype=torch.float64, device=dev)
    cdf = torch.cumsum(w / w.sum(), 0)
    i = torch.searchsorted(cdf, torch.rand(n, generator=g, device=dev, dtype=torch.float64)).clamp_(max=items - 1).to(torch.int32)
    if os.environ.get("UNIFORM"):
        i = torch.randint(0, items, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    j = torch.randint(0, items, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    nw = 64 * batch
    st.train_epoch(u[:nw], i[:nw], j[:nw], batch)
    torch.cuda.synchronize()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for rep in range(3):
        assert lib.crec_debug_stamps_clear() == 0
        st.train_epoch(u[nw:], i[nw:], j[nw:], batch)
        torch.cuda.synchronize()
        nbuf = 8 * 16384
        buf = (ctypes.c_ulonglong * nbuf)()
        assert lib.crec_debug_stamps(buf, ctypes.c_size_t(nbuf)) == 0
        a = np.frombuffer(buf, dtype=np.uint64)[8 * 16000:8 * 16000 + 16 * W].reshape(W, 16).astype(np.int64)
        mins = (0, 2, 9)
        t = a.astype(np.float64)
        for k in mins:
            t[:, k] = np.where(a[:, k] > 0, (1 << 62) - a[:, k], 0)
        t0 = t[0, 0]
        us = np.where(t > 0, (t - t0) / 100.0, np.nan)
        print("rep %d   (us from the launch's first wave; 100 MHz clock)" % rep)
        print("batch  start  wait_end(min max)  loads1_max  tickets(min max)  chunk_fin  fin_start  fin_sum  fin_stamp  last_stamp  end")
        for b in range(W):
            r = us[b]
            print("%5d %6.2f   %6.2f %6.2f     %6.2f      %6.2f %6.2f     %6.2f    %6.2f   %6.2f    %6.2f     %6.2f  %6.2f"
                  % (b, r[0], r[2], r[3], r[4], r[9], r[10], r[11], r[5], r[6], r[7], r[8], r[1]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
