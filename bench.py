#!/usr/bin/env python3
"""bench.py -- BPR triples/sec on the ml-1m-shaped workload (BASELINE.json configs[1]).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one batch (6144 triples) through the hot path: the fused gather -> BPR loss ->
per-row gradient sums -> TF1-semantics Adam (bpr_step_fused, one launch per batch), including the
per-epoch-chunk index build and the loss reduction that the library launches around them.
The sampled index stream (bit-exact host sampler, seed 1) and the tables are resident in HBM
when the timed region starts.  Prints ONE JSON line (rank 0).

Extra objects in the line:
  roofline      dominant kernel (bpr_step_fused): algorithmic bytes per launch / its average
                duration from HIP events on the launch stream, against the 8 TB/s HBM peak
  cpu_baseline  the CPU oracle (oracle/crec_oracle.c, one core) on a bounded sample of the
                same workload -- a reported baseline, not the target
  eval          full-catalogue top-20 evaluation of all test users (users/s)
  sampler       the bit-exact device sampler's rate and the measured sampler-inclusive epoch rate (default mode), the host form beside it
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
D, B, NEG, LR, REG = 64, 6144, 4, 1e-3, 0.01


def launch_ranks(n, argv=None):
    """`python bench.py --gpus N` without a launcher around it: start N copies of this script, one per GPU, with the
    environment `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` would give them (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT = a free port), relay rank 0's JSON line and return non-zero if
    any rank failed (the others are terminated: a dead peer would leave them blocked in a collective)."""
    import socket
    import subprocess
    argv = list(sys.argv[1:] if argv is None else argv)
    def free_port():
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            return str(so.getsockname()[1])
    port = os.environ.get("MASTER_PORT") or free_port()
    # the transport canaries form process groups of their own before the ranks': their ports are probed here like MASTER_PORT
    # (a rank started by torch.distributed.run has no launcher of ours and falls back to MASTER_PORT + 1 / + 2)
    canary_ports = os.environ.get("CREC_CANARY_PORTS") or (free_port() + "," + free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0",
                   CREC_CANARY_PORTS=canary_ports)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    got = []
    reader = threading.Thread(target=lambda: got.append(procs[0].stdout.read()), daemon=True)  # rank 0 prints one line; EOF at exit
    reader.start()
    rc, live = 0, set(range(n))
    while live:  # a rank that dies takes the run with it: its peers would sit in the rendezvous or a collective for ever
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0:
                sys.stderr.write("bench.py: rank %d exited with %d\n" % (r, code))
                if rc == 0:
                    rc = code if code > 0 else 1
                    for q in procs:  # exact children of this process, by handle
                        if q.poll() is None:
                            q.terminate()
        time.sleep(0.05)
    reader.join(timeout=10)
    line0 = got[0] if got else b""
    sys.stdout.buffer.write(line0)
    sys.stdout.flush()
    return rc


def dry_rank(real_stdout):
    """--dry-launch: what a rank does before it needs a GPU -- read the launcher's environment, join the process group
    (gloo: the CPU stand-in for RCCL), agree with the peers.  Rank 0 prints one JSON line."""
    import torch
    import torch.distributed as dist
    world, rank, local_rank = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    seen = None
    if os.environ.get("CREC_DRY_FAIL_RANK") == str(rank):  # test hook: a rank that dies before the rendezvous
        return 3
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.zeros(world, dtype=torch.int64)
        t[rank] = 1 + local_rank
        dist.all_reduce(t)
        seen = t.tolist()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        os.write(real_stdout, (json.dumps({"dry_launch": True, "world": world, "ranks_seen": seen,
                                           "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"))}) + "\n").encode())
    return 0


def main():
    # the host driver only supports dmabuf IPC: RCCL between processes needs this before HIP is initialised
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # stdout carries exactly one line, the JSON: everything else that lands on fd 1 (RCCL prints a version banner
    # there when it initialises) is sent to stderr
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("CREC_BENCH_TRACE_AFTER"):  # diagnosis of a run that does not come back: every thread's Python stack to stderr after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["CREC_BENCH_TRACE_AFTER"]), repeat=False, file=sys.stderr)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=45600)
    ap.add_argument("--warmup", type=int, default=65536)
    ap.add_argument("--cpu-steps", type=int, default=3000, help="steps of the scalar-C CPU baseline sample (0 = skip all CPU legs); 3000 ~ 7 s")
    ap.add_argument("--no-extras", action="store_true", help="skip the roofline/eval/sampler/cpu legs")
    ap.add_argument("--no-big", action="store_true", help="skip the S-big (10M x 1M, d=128) leg")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher rehearsal without a GPU: every rank joins a gloo group, all-reduces its rank and reports its environment")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started bare (`python bench.py --gpus N`): this process becomes the launcher.  It has not imported torch or
        # touched HIP and never will -- a process that has initialised the GPU must not be replaced or forked into ranks.
        os.dup2(real_stdout, 1)
        os.close(real_stdout)
        sys.exit(launch_ranks(a.gpus))

    if a.dry_launch:
        sys.exit(dry_rank(real_stdout))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("CREC_BENCH_ONE_GPU") == "1" and world > 1:  # (the rehearsal on one GPU, below: every rank -- and its canary -- on cuda:0)
        os.environ["LOCAL_RANK"] = "0"
        local_rank = 0
        # a rank's exchange kernel waits inside the GPU for its peers' kernels.  On its own GPU it may fill the chip while it waits (the
        # fused serve / apply exchanges of the row-sharded step launch up to 1024 workgroups); on ONE shared GPU that starves the very
        # kernels it waits for.  The five-launch form keeps every waiting launch at 128 workgroups.
        os.environ.setdefault("CREC_SHARD_UNFUSED", "1")
    # N > 1: which transport carries the exchanges (CREC_TRANSPORT = auto | peer | rccl).  The peer transport (IPC windows, one
    # kernel per exchange, csrc/crec_peer.hip) is the fast one, but it reads other GPUs' memory from kernels of this process: it is
    # first proven between THESE GPUs by a canary in child processes (tools/micro/peer_canary.py: every exchange bit for bit, then
    # timings) -- started before this process touches the GPU.  Only if every rank's canary passed does the run use it; else RCCL.
    want_transport = os.environ.get("CREC_TRANSPORT", "auto")
    canary = None
    t_legs = {}  # seconds per leg of this run (rank 0's), printed with the line
    t_leg0 = time.perf_counter()
    if world > 1 and want_transport in ("auto", "peer"):
        import subprocess
        base_port = int(os.environ.get("MASTER_PORT", "29511"))
        ports = [int(x) for x in os.environ.get("CREC_CANARY_PORTS", "%d,%d" % (base_port + 1, base_port + 2)).split(",")]
        env = dict(os.environ, MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"))
        # started by torch.distributed.run, this rank is told to find its store at the launcher's agent (MASTER_PORT): the canaries
        # form a group of their OWN on ports of their own, whose rank 0 must host its store itself
        for k in [k for k in env if k.startswith("TORCHELASTIC_") or k.startswith("TORCH_NCCL_ASYNC")]:
            env.pop(k)
        # a form gets CREC_CANARY_TIMEOUT seconds (default 45): two hung forms cost 90 s, not the 600 s a whole bench run may be given
        canary_s = float(os.environ.get("CREC_CANARY_TIMEOUT", "45"))
        for form in ("0", "1"):  # written-through system-scope accesses first, then the fence form (csrc/crec_peer.hip)
            try:
                if os.environ.get("CREC_CANARY_FORCE_FAIL") == "1":  # test hook: a node whose GPUs cannot map each other's windows
                    raise RuntimeError("forced failure (CREC_CANARY_FORCE_FAIL=1)")
                cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "micro", "peer_canary.py")], env=dict(env, CREC_PEER_FENCES=form, MASTER_PORT=str(ports[int(form)])),
                                    stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=canary_s)
                canary = {"rc": cp.returncode, "fences": form == "1"}
                lines = [ln for ln in cp.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
                if lines:
                    canary.update(json.loads(lines[-1]))
                if cp.returncode != 0:
                    canary["stderr_tail"] = cp.stderr.decode(errors="replace")[-400:]
            except Exception as e:  # noqa: BLE001  (a canary that hangs or cannot start: RCCL it is)
                canary = {"rc": -1, "error": repr(e), "fences": form == "1"}
            # every rank's canary belongs to one group and rank 0 of it gathers the verdict: all ranks see the same exit code
            if canary["rc"] == 0:
                os.environ["CREC_PEER_FENCES"] = form
                break
            print("bench.py rank %d: the peer transport's canary (%s form) failed: %r" % (rank, "fence" if form == "1" else "write-through", canary),
                  file=sys.stderr, flush=True)
        if canary["rc"] != 0 and want_transport == "peer":
            print("bench.py rank %d: CREC_TRANSPORT=peer was asked for and its canary failed" % rank, file=sys.stderr, flush=True)
            if rank == 0:
                os.write(real_stdout, (json.dumps({"metric": "BPR triples/sec (train)", "value": None, "n_gpus": world, "error": "peer transport canary failed",
                                                   "peer_canary": canary}) + "\n").encode())
            sys.exit(4)
    t_legs["canary"] = time.perf_counter() - t_leg0

    out = None
    try:  # ---- the headline: whatever fails in here is reported in rank 0's line and by the exit code
        import torch
        import torch.distributed as dist
        from cleverrec_amd import runtime
        from cleverrec_amd.data import synthetic

        force_dp = os.environ.get("CREC_FORCE_DP") == "1"  # rehearse the N > 1 code path on one GPU
        # CREC_BENCH_ONE_GPU=1: a rehearsal of `--gpus N` on ONE GPU -- every rank on cuda:0, the control plane over gloo (RCCL refuses two
        # ranks on one device), the data over the peer windows: the whole N > 1 flow of this file between real processes, minus the links
        one_gpu = os.environ.get("CREC_BENCH_ONE_GPU") == "1" and world > 1
        if one_gpu:
            local_rank = 0
        ctl_dev = "cpu" if one_gpu else None  # where the control plane's little tensors live (None: this rank's GPU)
        if world > 1 or force_dp:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            if one_gpu:
                dist.init_process_group("gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        assert world == a.gpus, "--gpus %d but WORLD_SIZE=%d" % (a.gpus, world)

        sess = runtime.Session(local_rank)
        # N > 1 (weak scaling): users are sharded, every rank holds one ml-1m-shaped shard of users (its own
        # P rows, its own sampled stream of 6144 triples per step); Q is replicated and kept identical by one
        # RCCL all-reduce of the per-item gradient sums per step (cleverrec_amd/parallel.py).
        # every rank draws the SAME synthetic population (one catalogue by construction: the replicas of Q line up); what differs
        # per rank is the sampled index stream (sampler seed 1 + rank) and its own P rows
        data = synthetic("S-ml1m", seed=20240601)
        P0 = torch.empty(data.user_nums, D).normal_(0, 0.01, generator=torch.Generator().manual_seed(1000 + rank)).numpy()
        Q0 = torch.empty(3706, D).normal_(0, 0.01, generator=torch.Generator().manual_seed(0)).numpy()[:data.item_nums]
        dp = world > 1 or force_dp
        transport = "rccl"
        if dp:
            from cleverrec_amd.parallel import DpBprState, agree_min
            assert agree_min(data.item_nums) == data.item_nums == -agree_min(-data.item_nums), "replicas need one catalogue"
            if world > 1 and want_transport != "rccl" and agree_min(1 if canary and canary.get("rc") == 0 else 0) == 1:
                transport = "peer"
            elif one_gpu:
                raise RuntimeError("CREC_BENCH_ONE_GPU=1 needs the peer transport (its canary failed: %r)" % (canary,))
            elif os.environ.get("CREC_TRANSPORT") == "peer" and world == 1:
                transport = "peer"  # (one rank: rehearses the code path, like CREC_FORCE_DP)
            # (the window also carries the S-big leg's compact tables: 65536 triples per rank x 2 items x 512 B per half)
            st = DpBprState(sess, P0, Q0, lr=LR, reg=REG, transport=transport, window_bytes=288 << 20)
            if transport == "peer":
                # a rank's exchange kernel gives up on a peer after this long (then the run fails with the transport's error instead of
                # hanging): the ranks of a bench are only loosely in step -- one may still be filling the S-big leg's tables (GBs of
                # normal_()) when another already waits for it -- so well above the library's 20 s
                sess.ctx.check(sess.lib.crec_comm_set_timeout(sess.ctx.h, 120.0))
        else:
            st = runtime.BprState(sess, P0, Q0, lr=LR, reg=REG, adam_mode="tf1")

        sess.ctx.seed(1 + rank)
        t0 = time.perf_counter()
        u, i, j = sess.ctx.sample_pairwise(data.train_csr, data.item_nums, NEG)
        t_sample = time.perf_counter() - t0
        n_epoch = len(u)
        if dp:
            n_epoch = agree_min(n_epoch // B) * B  # the dp epoch is collective: same batch count everywhere
            u, i, j = u[:n_epoch], i[:n_epoch], j[:n_epoch]
        du, di, dj = (sess.to_dev(x) for x in (u, i, j))

        views = {}

        def run_steps(k):
            """k consecutive batches, wrapping around the sampled epoch."""
            done = 0
            nb_epoch = n_epoch // B  # full batches only, so every step is 6144 triples
            while done < k:
                take = min(k - done, nb_epoch)
                if take not in views:  # input views and the loss buffer of a call shape are made once, outside any timed region
                    views[take] = (du[:take * B], di[:take * B], dj[:take * B], sess.empty((take,), torch.float32))
                vu, vi, vj, vl = views[take]
                st.train_epoch(vu, vi, vj, B, out=vl)
                done += take

        def fence():
            torch.cuda.synchronize()
            if dp:
                dist.barrier()
            torch.cuda.synchronize()

        run_steps(a.warmup)
        # Nothing may allocate inside the timed region: the library sizes its per-chunk workspaces (index, loss terms,
        # page-locked count buffer) by the number of batches of a call and only ever grows them, so one untimed pass
        # with exactly the calls the timed region will make leaves every buffer at its final size.  Then a time-based
        # pre-warm (clocks, caches, allocator; and a second of continuous launches a device-side utilisation sampler can see): at least 1 s of the same calls.  Both are reported.
        t_pre = time.perf_counter()
        prewarm_steps = 0
        while True:
            run_steps(a.steps)
            fence()
            prewarm_steps += a.steps
            more = time.perf_counter() - t_pre < 1.0
            if dp:  # the passes are collective: every rank takes rank 0's decision
                flag = torch.tensor([int(more)], dtype=torch.int64, device=ctl_dev or sess.device)
                dist.broadcast(flag, src=0)
                more = bool(flag.item())
            if not more:
                break
        prewarm_s = time.perf_counter() - t_pre
        t0 = time.perf_counter()
        run_steps(a.steps)
        fence()
        dt = time.perf_counter() - t0
        if dp:
            st.synchronize()  # (the epochs report an exchange that gave up on a peer late, without draining the stream: ask now)
            tt = torch.tensor([dt], dtype=torch.float64, device=ctl_dev or sess.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        value = a.steps * B * world / dt

        # A call has fixed costs (the chunk's index kernel, one read-back of its launch sizes, the loss reduction: ~0.1 ms)
        # that a timed region of a few steps does not amortise.  When --steps is small the line also carries the rate of
        # whole-epoch calls (the shape train_model() uses: one call per epoch of ~455 batches), measured right after.
        steady = None
        nb_epoch_ = n_epoch // B
        if a.steps < 4 * nb_epoch_:
            k_st = 0
            t0 = time.perf_counter()
            while True:
                run_steps(nb_epoch_)
                k_st += nb_epoch_
                more = k_st < 8 * nb_epoch_
                if dp:
                    flag = torch.tensor([int(more)], dtype=torch.int64, device=ctl_dev or sess.device)
                    dist.broadcast(flag, src=0)
                    more = bool(flag.item())
                if not more:
                    break
            fence()
            dts = time.perf_counter() - t0
            if dp:
                tt = torch.tensor([dts], dtype=torch.float64, device=ctl_dev or sess.device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dts = float(tt.item())
            steady = {"steps": k_st, "ms_per_step": dts / k_st * 1e3, "value": k_st * B * world / dts, "unit": "triples/s",
                      "note": "calls of one whole epoch (%d batches) each, the shape train_model() uses; `value` above is exactly --steps "
                              "steps in one call and carries that call's fixed costs" % nb_epoch_}

        out = {
            "metric": "BPR triples/sec (train)", "value": value, "unit": "triples/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "prewarm_s": prewarm_s, "prewarm_steps": prewarm_steps, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "steady_state": steady,
            "config": {"workload": "BPR on S-ml1m (6040x3706, 1,000,208 interactions, ml-1m marginals), d=64 fp32, "
                                   "batch 6144, neg_ratio 4, Adam tf1 semantics (BASELINE.json configs[1])",
                       "users": data.user_nums, "items": data.item_nums, "embed_size": D, "batch_size": B,
                       "neg_ratio": NEG, "adam_mode": "tf1", "step_kernel": "bpr_step_fused (a launch per batch)" if dp else "bpr_steps_flow (up to 64 batches per launch) where resident, else bpr_step_fused", "sharding": ("users sharded (one S-ml1m user shard per GPU), Q replicated, 1 all-reduce of dQ per step over %s"
                                                         % ("the peer windows (xGMI loads from one kernel per exchange)" if transport == "peer" else "RCCL")) if dp else "none"},
        }
        if dp:
            out["transport"] = transport
            if canary is not None:
                out["peer_canary"] = canary
    except Exception as e:  # noqa: BLE001
        import traceback
        traceback.print_exc()
        if rank == 0:
            os.write(real_stdout, (json.dumps({"metric": "BPR triples/sec (train)", "value": None, "unit": "triples/s", "n_gpus": world, "steps": a.steps,
                                               "warmup": a.warmup, "error": "the headline run failed: %r" % (e,), "peer_canary": canary, "leg_seconds": t_legs}) + "\n").encode())
        sys.exit(1)
    t_legs["headline"] = time.perf_counter() - t_leg0 - t_legs["canary"]
    leg_errors = {}  # legs behind the headline: a failure is recorded in the line, the run goes on, the exit code stays 0

    prof = None
    t_leg = time.perf_counter()
    if not a.no_extras:
        try:
            # ---- roofline: per-kernel durations from the HIP events attached to each dispatch.  The data-parallel
            # epoch is collective, so at N > 1 EVERY rank runs these steps; rank 0 reports its own kernels.
            # The pass is the timed region's workload again, long enough for a stable mean whatever --steps was (a 20-step
            # pass is ONE call: 19 launches right after the event pool is first filled read 9 us where thousands read 6.3);
            # a first pass fills the pool and is discarded.
            k_prof = max(a.steps, 4 * (n_epoch // B))
            sess.ctx.profile(True)
            run_steps(min(k_prof, n_epoch // B))
            fence()
            sess.ctx.profile(True)  # resets the sums, keeps the events
            run_steps(k_prof)
            fence()
            prof = sess.ctx.profile_read()
            flow_steps = sess.ctx.profile_units("bpr_steps_flow")
            sess.ctx.profile(False)
        except Exception as e:  # noqa: BLE001
            prof = None
            leg_errors["kernel_profile"] = repr(e)
            print("bench.py rank %d: the per-kernel profiling pass failed: %r" % (rank, e), file=sys.stderr, flush=True)
    t_legs["kernel_profile"] = time.perf_counter() - t_leg
    if rank == 0 and prof is not None:
        U, I = data.user_nums, data.item_nums
        # algorithmic bytes per launch (DESIGN.md "Algorithmic bytes"): SURVEY 8(d)'s 24d+12 per triple
        # split by which kernel owns the access, plus the tf1 Adam sweep 24*d*rows
        alg = {"bpr_step_fused": B * (24 * D + 12) + 24 * D * (U + I)}
        kern = {}
        for name, (ms, n) in prof.items():
            if n:
                kern[name] = {"avg_us": ms / n * 1e3, "launches": n}
        # the tf1 step runs as ONE dataflow launch per up to 64 batches where every wave of it is resident (bpr_steps_flow, the
        # default at this shape), else as a launch per batch (bpr_step_fused; always for the data-parallel step)
        flow = prof.get("bpr_steps_flow", (0, 0))[1] > 0
        dom = "bpr_steps_flow" if flow else "bpr_step_fused"
        steps_per_launch = flow_steps / prof[dom][1] if flow else 1.0
        alg_step = alg["bpr_step_fused"]
        # HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes over this same
        # command, tools/profile.sh; gfx950 FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes).  PMC
        # counters cannot be read from inside the process, so the committed measurement is quoted.
        traffic = None
        def newest(name):  # the latest round's committed summary of this workload
            for r in ("round4", "round3", "round2", "round1"):
                f_ = os.path.join(ROOT, "profiles", "%s_%s" % (r, name))
                if os.path.exists(f_):
                    return f_
            return os.path.join(ROOT, "profiles", "round4_" + name)
        tf = newest("traffic.json")
        if os.path.exists(tf) and not dp:
            tj = json.load(open(tf))
            if tj.get("kernel", "bpr_step_fused") == dom:
                traffic = tj.get("traffic_bytes_per_launch")
        if dp:
            alg_step = B * (24 * D + 12) + 24 * D * U + 4 * D * I  # P swept, dQ written; Q is swept by bpr_adam_dense / peer_allreduce_adam
        alg[dom] = alg_step * steps_per_launch  # algorithmic bytes of an average launch (the launches of an epoch hold 64 .. 7 steps)
        avg_s = prof[dom][0] / prof[dom][1] * 1e-3
        ach = alg[dom] / avg_s / 1e9
        # the second clock: rocprofv3 --kernel-trace --stats of this command (tools/profile.sh -> profiles/roundN_kernel_stats.txt), the
        # same mix of launches.  `frac` is priced on the SLOWER of the two clocks; both are reported, and `clock` says which it was
        rp_us = None
        ks = newest("kernel_stats.txt")
        if os.path.exists(ks) and not dp:
            for line in open(ks):
                f = line.split()
                if line.startswith(dom) and len(f) >= 5:
                    try:
                        rp_us = float(f[-2]) / 1e3
                    except ValueError:
                        pass
                    break
        slow_us = max(avg_s * 1e6, rp_us or 0.0)
        out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": alg[dom] / slow_us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": alg[dom] / slow_us / 1e3 / HBM_PEAK_GBS, "traffic": traffic, "algorithmic_bytes_per_launch": alg[dom],
                           "algorithmic_bytes_per_step": alg_step, "steps_per_launch": steps_per_launch,
                           "avg_launch_us": avg_s * 1e6, "avg_us_per_step": avg_s * 1e6 / steps_per_launch,
                           "avg_launch_us_rocprof": rp_us, "frac_hip_events": ach / HBM_PEAK_GBS,
                           "clock": ("rocprofv3 --kernel-trace --stats, committed: " + os.path.relpath(ks, ROOT)) if (rp_us or 0.0) > avg_s * 1e6
                                    else "HIP events attached to each dispatch, measured in this run",
                           "note": "avg_launch_us = mean over the timed launches of the elapsed time between the two HIP events "
                                   "attached to each dispatch (hipExtLaunchKernelGGL start/stop events on the launch stream), i.e. "
                                   "the dispatch's own begin/end timestamps; avg_launch_us_rocprof = the committed rocprofv3 "
                                   "--kernel-trace --stats average of the same workload (" + os.path.relpath(ks, ROOT) + "); achieved and "
                                   "frac use the larger of the two.  A bpr_steps_flow launch holds up to 64 steps: its algorithmic bytes "
                                   "are steps_per_launch x the per-step count.  Tables+slots are 7.5 MB (Infinity-Cache resident): the HBM "
                                   "fraction is nominal; S-big (synthetic_big) is the HBM-resident form",
                           "launch_period_us": None if flow else dt / a.steps * 1e6 - sum(v["avg_us"] * v["launches"] for k, v in kern.items()
                                                                                          if k != dom) / max(kern[dom]["launches"], 1)}
        step_bytes = alg_step
        out["kernels"] = kern
        out["step_algorithmic_GBs"] = step_bytes / (dt / a.steps) / 1e9

    big_sharded = None
    t_leg = time.perf_counter()
    if not a.no_extras and not a.no_big and dp:
        # ---- S-big row-sharded over the N ranks (BASELINE configs[4]; collective: every rank runs it) ----------
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_big

            def big_max(x):
                tt = torch.tensor([x], dtype=torch.float64, device=ctl_dev or sess.device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                return float(tt.item())
            # (the one-GPU rehearsal: two processes' kernels that wait for each other take turns on the one GPU -- a hand-shake costs a
            # scheduling turn there, so it runs a handful of steps)
            big_sharded = bench_big.run_sharded(steps=4 if one_gpu else 200, sess=sess, rank=rank, world=world, barrier=fence, max_reduce=big_max)
            # the same with 65536 triples per rank and step (batches beyond 8192 are indexed in global memory, csrc/index_big.hip):
            # the exchanges' hand-shakes and the launches are paid once per ten times the triples
            big_sharded["larger_batch"] = bench_big.run_sharded(steps=2 if one_gpu else 24, batch=65536, sess=sess, rank=rank, world=world, barrier=fence,
                                                                max_reduce=big_max)
        except Exception as e:  # every rank fails or none: the sizes are the same everywhere
            big_sharded = {"error": repr(e)}
            print("rank %d: the S-big row-sharded leg failed: %r" % (rank, e), file=sys.stderr, flush=True)
    if rank == 0 and big_sharded is not None:
        out["synthetic_big"] = big_sharded
        t_legs["synthetic_big_sharded"] = time.perf_counter() - t_leg

    if rank == 0 and prof is not None and world == 1 and not force_dp:  # single-process legs only at N = 1
        U, I = data.user_nums, data.item_nums
        t_leg = time.perf_counter()
        try:
            # ---- eval: all test users, full catalogue, K = 20 ---------------------------------------
            users = sess.to_dev(data.test_users)
            sp, si = sess.to_dev(data.seen_csr[0]), sess.to_dev(data.seen_csr[1])
            tp, ti = sess.to_dev(data.test_csr[0]), sess.to_dev(data.test_csr[1])

            def eval_once():
                ids, _ = runtime.eval_topk(sess, st.P, st.Q, users, sp, si, 20)
                return runtime.rank_metrics(sess, ids, users, [10, 20], tp, ti)
            eval_once()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                hr, mrr, ndcg = eval_once()
            torch.cuda.synchronize()
            te = (time.perf_counter() - t0) / reps
            out["eval"] = {"metric": "eval users/sec@top-20 (full catalogue + HR/MRR/NDCG@10,20)", "value": len(data.test_users) / te,
                           "unit": "users/s", "ms": te * 1e3, "users": int(len(data.test_users)), "items": I,
                           "gemm_TFLOPs": 2.0 * len(data.test_users) * I * D / te / 1e12, "ndcg@10": float(ndcg[0].mean())}

        except Exception as e:  # noqa: BLE001
            leg_errors['eval'] = repr(e)
            print("bench.py rank %d: the eval leg failed: %r" % (rank, e), file=sys.stderr, flush=True)
        t_legs['eval'] = time.perf_counter() - t_leg
        t_leg = time.perf_counter()
        try:
            # ---- sampler ---------------------------------------------------------------------------------
            # the default of the drop-in classes: the reference's own index stream (NumPy MT19937, bit for bit), drawn ON the
            # device (runtime.MtSampler, csrc/mt_sampler.hip), the next epoch's sample enqueued ahead of this epoch's steps (same stream)
            epoch_gpu_s = (n_epoch / B) * (dt / a.steps)
            ms_dev = runtime.MtSampler(sess, data)
            key0, pos0 = np.random.RandomState(1).get_state()[1:3]
            t_dev = []
            kk, pp = key0, pos0
            for _ in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                cols, (kk, pp) = ms_dev.sample("pairwise", NEG, kk, pp)
                t_dev.append(time.perf_counter() - t0)
            t_dev_s = sorted(t_dev[1:])[len(t_dev[1:]) // 2]
            same = bool(np.array_equal(ms_dev.sample("pairwise", NEG, key0, pos0)[0][2][:4096].cpu().numpy(), j[:4096])) if a.gpus == 1 else None
            np.random.seed(1)
            t_ep = []
            for e in range(7):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                cu, ci, cj = ms_dev.next_epoch("pairwise", NEG, ahead=True)
                st.train_epoch(cu, ci, cj, B)
                torch.cuda.synchronize()
                t_ep.append(time.perf_counter() - t0)
            ms_dev.drop()
            steady_ep = sorted(t_ep[1:])[len(t_ep[1:]) // 2]
            stats_dev = ms_dev.stats()
            out["sampler"] = {"mode": "mt19937 bit-exact, drawn on the device (jump-ahead MT19937 chains, randint as a compaction, negatives / shuffle-target "
                                      "walks, permutation by sort + pointer chase); the default of the model classes",
                              "triples_per_s": int(cu.numel()) / t_dev_s, "ms_per_epoch": t_dev_s * 1e3, "epoch_triples": int(cu.numel()),
                              "epoch_inclusive_triples_per_s": int(cu.numel()) / steady_ep,
                              "epoch_inclusive_note": "measured: median of 6 epochs of next_epoch() + %d steps, the next epoch's sample enqueued in front of the "
                                                      "steps on the same stream (the host never waits for it); first epoch %.1f ms, steady %.1f ms; the sampler alone %.1f ms, the "
                                                      "steps alone %.1f ms" % (int(cu.numel()) // B, t_ep[0] * 1e3, steady_ep * 1e3, t_dev_s * 1e3, epoch_gpu_s * 1e3),
                              "epoch_inclusive_triples_per_s_serial": int(cu.numel()) / (t_dev_s + epoch_gpu_s),
                              "first_4096_negatives_equal_host_sampler": same,
                              "walks": {k: stats_dev[k] for k in ("neg_tiles", "neg_us", "perm_tiles", "perm_us", "words_negatives", "words_shuffle")}}
            ms_dev.close()

            # the host form of the same stream (sampler_mode=mt19937_host; SBPR's sampler and catalogues beyond the device form's limits)
            out["sampler"]["host_form"] = {"triples_per_s": n_epoch / t_sample, "ms_per_epoch": t_sample * 1e3,
                                           "note": "crec_sample_pairwise: one RNG thread + swap and gather threads behind it"}
            from cleverrec_amd.utils import sampler as S
            S.drop_prefetch()
            np.random.seed(1)
            keep, t_ep = [], []
            for e in range(6):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                nb_, uu, ii, jj = S.pairwise_ranking_sampler(data, NEG, B, prefetch_next=True)
                keep = [uu, ii, jj]
                st.train_epoch(sess.to_dev(uu), sess.to_dev(ii), sess.to_dev(jj), B)
                torch.cuda.synchronize()
                t_ep.append(time.perf_counter() - t0)
            S.drop_prefetch(join=True)
            steady = sorted(t_ep[1:])[len(t_ep[1:]) // 2]
            out["sampler"]["host_form"]["epoch_inclusive_triples_per_s"] = len(uu) / steady
            out["sampler"]["host_form"]["epoch_inclusive_note"] = ("median of 5 epochs (sample -> upload -> steps), next epoch's sample prefetched on a host thread; "
                                                                   "first epoch %.1f ms, steady %.1f ms" % (t_ep[0] * 1e3, steady * 1e3))

            # the optional device sampler (same acceptance rule, counter-based generator: not NumPy-stream compatible)
            ds = runtime.DeviceSampler(sess, data, seed=1)
            ds.pairwise(NEG)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                cu, ci, cj = ds.pairwise(NEG)
            torch.cuda.synchronize()
            t_dev = (time.perf_counter() - t0) / 5
            out["sampler"]["counter_based_mode"] = {"triples_per_s": int(cu.numel()) / t_dev, "ms_per_epoch": t_dev * 1e3,
                                                    "epoch_inclusive_triples_per_s": n_epoch / (t_dev + epoch_gpu_s),
                                                    "note": "crec_sample_negatives_device + device randperm + gathers: same acceptance rule, own generator, NOT the "
                                                            "reference's stream; opt-in (sampler_mode=device)"}

        except Exception as e:  # noqa: BLE001
            leg_errors['sampler'] = repr(e)
            print("bench.py rank %d: the sampler leg failed: %r" % (rank, e), file=sys.stderr, flush=True)
        t_legs['sampler'] = time.perf_counter() - t_leg
        # ---- S-big: the HBM-bound form of the same path (10 M x 1 M, d=128, lazy Adam) ----------------
        if not a.no_big:
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_big
                out["synthetic_big"] = bench_big.run(steps=200, sess=sess)
                # the same step on larger batches (batch_size is the reference's .properties key, any int: model/Recommender.py:17;
                # beyond 8192 the batch's index is built in global memory, csrc/index_big.hip): a launch long enough that the
                # dispatch and the multi-occurrence rows' tails no longer show
                out["synthetic_big"]["larger_batches"] = {}
                for bb in (32768, 65536):
                    r = bench_big.run(steps=(1 << 23) // bb, warmup=(1 << 21) // bb, batch=bb, sess=sess)
                    out["synthetic_big"]["larger_batches"][str(bb)] = {k: r[k] for k in ("triples_per_s", "us_per_step", "bpr_triples_lazy",
                                                                                          "unique_user_rows_per_batch", "unique_item_rows_per_batch")}
                # HBM-side bytes per launch of the step kernel from the committed PMC passes over the same commands (tools/profile.sh:
                # FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes, MI355X_MICROARCH.md's gfx950 correction) -> a roofline object per batch size
                def lazy_traffic(name):
                    f_ = newest(name)
                    vals = {}
                    if os.path.exists(f_):
                        for line in open(f_):
                            t = line.split()
                            if line.startswith("bpr_triples_lazy") and len(t) >= 5:
                                vals[t[1]] = float(t[3])
                    return int((2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024) if {"FETCH_SIZE", "WRITE_SIZE"} <= set(vals) else None

                def lazy_roofline(k, traffic):
                    # (option lazy_window: a launch holds batches_per_launch batches; bytes, traffic and time are per batch here, and
                    # frac_hbm_side prices what the counters saw instead of the algorithmic count)
                    bpl = k.get("batches_per_launch", 1.0)
                    per_batch = None if traffic is None else int(traffic / bpl)
                    return {"bound": "hbm", "kernel": "bpr_triples_lazy", "achieved": k["GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": k["frac_of_8TBs"], "traffic": per_batch, "algorithmic_bytes_per_launch": k["algorithmic_bytes"] * bpl,
                            "algorithmic_bytes_per_batch": k["algorithmic_bytes"], "batches_per_launch": bpl,
                            "avg_launch_us": k["avg_us"] * bpl, "avg_us_per_batch": k["avg_us"],
                            "frac_hbm_side": None if per_batch is None else per_batch / (k["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS}
                out["synthetic_big"]["roofline"] = lazy_roofline(out["synthetic_big"]["bpr_triples_lazy"], lazy_traffic("big_pmc_summary.txt"))
                out["synthetic_big"]["larger_batches"]["65536"]["roofline"] = lazy_roofline(
                    out["synthetic_big"]["larger_batches"]["65536"]["bpr_triples_lazy"], lazy_traffic("bigbatch_pmc_summary.txt"))
                out["synthetic_big"]["sharded_world1"] = bench_big.run_sharded(steps=200, sess=sess)  # protocol cost
                import bench_eval_big
                out["eval_big"] = bench_eval_big.run(sess=sess, reps=2)  # 65536 users x 1 M items: MFMA utilisation
            except Exception as e:  # e.g. not enough free HBM on a shared box
                out["synthetic_big"] = {"error": repr(e)}

        t_leg = time.perf_counter()
        try:
            # ---- CPU baseline: the oracle on a bounded sample of the same workload --------------------
            if a.cpu_steps > 0:
                import oracle
                om = oracle.BprModel(P0, Q0, lr=LR, reg=REG, adam_mode=0)
                k, nbe = a.cpu_steps, n_epoch // B
                t0 = time.perf_counter()
                for s_ in range(k):  # the same sampled epoch, walked again when it runs out (the tables keep training)
                    b = s_ % nbe
                    om.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], j[b * B:(b + 1) * B])
                tc = time.perf_counter() - t0
                out["cpu_baseline"] = {"value": k * B / tc, "unit": "triples/s", "cores": 1, "kind": "port",
                                       "sample": "%d steps of 6144 triples over the same sampled epoch (%d batches, walked %.1f "
                                                 "times), oracle/crec_oracle.c orc_bpr_step (scalar C, tf1 Adam), %.1f s; host has "
                                                 "%d cores" % (k, nbe, k / nbe, tc, os.cpu_count())}
                # the reference-STYLE path of SURVEY 8(d): pure-Python sampler loop + vectorised torch-CPU fp32 step (tf1 Adam
                # sweep) at 1, 16 and (at most) 64 threads -- 64: the most a 6144-row operand keeps busy; with all 256 hardware threads of the GPU host one step took 9 s in round 2, and the argsort / Python-filter evaluation loop -- each on a bounded sample
                from oracle import reference_style
                n_all = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
                out["cpu_baseline"]["reference_style"] = reference_style.measure(data, P0, Q0, u, i, j, B, NEG, sorted({1, min(8, n_all), min(16, n_all), min(64, n_all)}), budget_s=3.0)
                out["cpu_baseline"]["reference_style"]["note"] = ("TensorFlow 1.x is not installable here: the TF1-CPU path is represented by the same "
                                                                  "structure on PyTorch-CPU (oracle/reference_style.py); cores = torch.set_num_threads(n)")
        except Exception as e:  # noqa: BLE001
            leg_errors['cpu_baseline'] = repr(e)
            print("bench.py rank %d: the cpu_baseline leg failed: %r" % (rank, e), file=sys.stderr, flush=True)
        t_legs['cpu_baseline'] = time.perf_counter() - t_leg
    if rank == 0:
        out["leg_seconds"] = {k: round(v, 2) for k, v in t_legs.items()}
        if leg_errors:
            out["leg_errors"] = leg_errors
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
