"""Where the prefetched epoch goes (GPU box): sampler into pageable vs page-locked arrays, on the main thread vs a worker."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cleverrec_amd import _lib, runtime
from cleverrec_amd.data import synthetic
def _pinned(n, dtype):
    return torch.empty(int(n), dtype={np.int32: torch.int32, np.float32: torch.float32}[dtype], pin_memory=True).numpy()
sess = runtime.Session(0)
data = synthetic("S-ml1m")
def once(alloc):
    c = _lib.Context(None); c.seed(1); c.alloc = alloc
    t0 = time.perf_counter(); out = c.sample_pairwise(data.train_csr, data.item_nums, 4); dt = time.perf_counter() - t0
    c.close(); return dt, out
for name, alloc in (("pageable", np.empty), ("pinned", _pinned), ("pageable", np.empty), ("pinned", _pinned)):
    dt, out = once(alloc); print("main thread, %s: %.1f ms" % (name, dt * 1e3))
res = {}
def work(alloc, key): res[key] = once(alloc)[0]
for name, alloc in (("pageable", np.empty), ("pinned", _pinned)):
    t = threading.Thread(target=work, args=(alloc, name)); t.start(); t.join(); print("worker thread, %s: %.1f ms" % (name, res[name] * 1e3))
dt, (u, i, j) = once(_pinned)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); a = [sess.to_dev(x) for x in (u, i, j)]; torch.cuda.synchronize(); print("upload pinned %.2f ms" % ((time.perf_counter() - t0) * 1e3))
dt, (u, i, j) = once(np.empty)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); a = [sess.to_dev(x) for x in (u, i, j)]; torch.cuda.synchronize(); print("upload pageable %.2f ms" % ((time.perf_counter() - t0) * 1e3))
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
