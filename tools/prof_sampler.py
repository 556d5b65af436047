"""Phases of the bit-exact host sampler on the GPU box's host (CREC_SAMPLER_TIMING=1 python tools/prof_sampler.py), with 0 .. 3
gather workers behind the shuffle and the gather's prefetch modes (CREC_SAMPLER_THREADS / CREC_SAMPLER_AHEAD are read once
per process: one process per setting)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
from cleverrec_amd import _lib
from cleverrec_amd.data import synthetic
data = synthetic("S-ml1m", seed=20240601)
ctx = _lib.Context(None)
ts = []
for rep in range(6):
    ctx.seed(1)
    t0 = time.perf_counter(); u, i, j = ctx.sample_pairwise(data.train_csr, data.item_nums, 4); ts.append(time.perf_counter() - t0)
print("median %%.1f ms for %%d triples" %% (sorted(ts)[len(ts) // 2] * 1e3, len(u)))
''' % ROOT
for ahead in os.environ.get("PROF_SAMPLER_AHEADS", "0 1 3").split():
    for th in ("0", "1", "2"):
        env = dict(os.environ, CREC_SAMPLER_THREADS=th, CREC_SAMPLER_TIMING="1", CREC_SAMPLER_AHEAD=ahead)
        if th not in os.environ.get("PROF_SAMPLER_WORKERS", "0 1 2").split():
            continue
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        lines = [l for l in r.stderr.splitlines() if "crec_sample_pairwise" in l or "negatives phase" in l]
        print("prefetch mode", ahead, "workers", th, "|", r.stdout.strip(), "|", lines[-1] if lines else r.stderr[-300:], flush=True)
        if os.environ.get("PROF_SAMPLER_ALL"):
            for l in lines:
                print("    ", l)
