#!/usr/bin/env python3
"""Per-wave timeline of the two lazy launches of an S-big step (debug build with -DCREC_STAMPS, tools/wave_stamps.sh
builds it): chunk waves (heavy rows) and row waves of the user side and of the item side."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    from cleverrec_amd import _lib
    import bench_big
    bench_big.run(steps=int(os.environ.get("STEPS", "40")), warmup=10)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    n = 8 * 16384
    buf = (ctypes.c_ulonglong * n)()
    assert lib.crec_debug_stamps(buf, ctypes.c_size_t(n)) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
    for name, part in (("user side", a[:8192]), ("item side", a[8192:])):
        live = (part[:, 6] > 0) & (part[:, 0] > 0)
        p = part[live]
        t0 = p[:, 0].min()
        start, end = (p[:, 0] - t0) * 10.0, (p[:, 1] - t0) * 10.0
        print("%s: %d waves, span %.2f us" % (name, len(p), end.max() / 1e3))
        q = [0, 10, 50, 90, 99, 100]
        for cls, cname in ((1, "chunk waves"), (2, "row waves")):
            m = p[:, 6] == cls
            if m.any():
                print("  %-11s %5d: start pct %s %s | duration pct %s | end max %.0f ns" % (
                    cname, m.sum(), q, np.percentile(start[m], q).round(0), np.percentile((end - start)[m], q).round(0), end[m].max()))


if __name__ == "__main__":
    main()
