"""Builds cleverrec_amd/lib/libcrec.so (gfx950 only) with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the .so travels
to the GPU box with the repo snapshot.  `python -m cleverrec_amd.build [--force] [--exact]`.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libcrec.so")

ARCH = "gfx950"
COMMON = ["-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per-file extra flags.  bpr_train keeps every fp32 mul/add separately rounded (TF1 / Eigen
# element-wise semantics; makes the Adam arithmetic bit-reproducible against the oracle).
SOURCES = {
    "crec_ctx.cpp": [],
    "crec_comm.cpp": ["-pthread"],
    "crec_peer.hip": ["-ffp-contract=off"],  # (its fused all-reduce + Adam repeats bpr_train.hip's arithmetic bit for bit)
    "shard_host.cpp": [],
    "crec_sampler.cpp": ["-fno-math-errno", "-pthread"],
    "gpu_sampler.hip": [],
    "mt_jump.cpp": ["-pthread"],
    "mt_sampler.hip": [],
    "bpr_train.hip": ["-ffp-contract=off"],
    "index_big.hip": [],
    "eval_topk.hip": [],
    "rows_adam.hip": ["-ffp-contract=off"],
    "ncf.hip": ["-ffp-contract=off"],
    "fm.hip": ["-ffp-contract=off"],
    "cml.hip": ["-ffp-contract=off"],
    "sbpr.hip": ["-ffp-contract=off"],
    "lrml.hip": ["-ffp-contract=off"],
    "transcf.hip": ["-ffp-contract=off"],
    "ncf_eval.hip": [],
}


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, exact=False):
    """exact: the -DCREC_EXACT_MATH variant (IEEE division / square root / expf in the optimizers' quotients and the sigmoid) into
    lib/exact/libcrec.so -- loaded with CREC_LIB=<that path>; the parity suite is run against it once per round (profiles/)."""
    libdir = os.path.join(LIBDIR, "exact") if exact else LIBDIR
    lib = os.path.join(libdir, "libcrec.so")
    os.makedirs(libdir, exist_ok=True)
    objdir = os.path.join(libdir, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "crec.h"))
    cc = hipcc()
    objs, jobs = [], []
    for src, extra in SOURCES.items():
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, [path] + headers + [os.path.abspath(__file__)]):
            jobs.append([cc, "--offload-arch=" + ARCH, "-x", "hip"] + COMMON + extra + (["-DCREC_EXACT_MATH"] if exact else []) + ["-c", path, "-o", obj])
    if jobs:  # translation units are independent: compile up to four at a time
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as pool:
            list(pool.map(run, jobs))
    if force or _stale(lib, objs):
        cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lrccl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, exact="--exact" in sys.argv))
