#!/usr/bin/env python3
"""lazy Adam step under option lazy_window at several table sizes: us/step over whole epochs for W = 0 (a launch per batch) and
W batches per launch.  Streams as tools/bench_big.py draws them (users uniform, positives Zipf(1.0) or uniform, negatives uniform)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import bench_big
    from cleverrec_amd import runtime
    sess = runtime.Session(0)
    shapes = [(6040, 3706, 64, 4096, True), (6040, 3706, 128, 6144, True), (100_000, 20_000, 64, 6144, True), (1_000_000, 100_000, 128, 6144, True),
              (1_000_000, 100_000, 128, 1024, True), (10_000_000, 1_000_000, 128, 6144, True), (10_000_000, 1_000_000, 128, 6144, False),
              (10_000_000, 1_000_000, 128, 2048, True), (10_000_000, 1_000_000, 32, 6144, True)]
    ws = [int(x) for x in os.environ.get("WS", "0 4 8 16").split()]
    for U, I, d, B, zipf in shapes:
        row = []
        for W in ws:
            sess.ctx.set_option("lazy_window", W)
            r = bench_big.run(U, I, d, steps=320, warmup=32, batch=B, sess=sess, zipf=zipf)
            row.append("W=%d %.2f" % (W, r["us_per_step"]))
        print("%9d x %8d d=%3d B=%4d %s: us/step  %s" % (U, I, d, B, "zipf" if zipf else "unif", "   ".join(row)), flush=True)


if __name__ == "__main__":
    main()
