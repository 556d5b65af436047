"""The peer transport (csrc/crec_peer.hip: IPC windows, one kernel per exchange) on ONE MI355X.

  * ranks as PROCESSES (tests/_peer_worker.py, world 2 and 3 on cuda:0): the real hipIpcGetMemHandle / hipIpcOpenMemHandle
    path, flags in uncached memory, system-scope fences -- random all-to-all-v rounds and all-reduces bit for bit, then the
    row-sharded epoch and the replicated-Q epoch of libcrec.so against the oracle, replicas bit-identical;
  * ranks as threads of this process (crec_comm_peer_connect_local): the same kernels through raw pointers, the sharded
    epoch against the oracle.
What one GPU cannot show is the xGMI side (peer memory types, link bandwidth): see DESIGN.md 7."""
import json
import os
import socket
import subprocess
import time
import sys
import threading

import numpy as np
import pytest
import torch

import oracle
from cleverrec_amd import parallel, runtime

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,fences,big", [(2, 0, 0), (3, 0, 0), (2, 1, 0), (2, 0, 1)])  # big: the epochs' index builds through csrc/index_big.hip
def test_ranks_as_processes_over_ipc_windows(world, fences, big):
    """(Not more ranks on ONE GPU: a rank's launch waits inside the GPU for its peers' launches, so all ranks' workgroups have to be
    resident together -- one per CU and rank; four ranks' worth do not fit a CU's registers, and the waits then run into their
    time-out.  One process per GPU, the production layout, has the CU to itself.)"""
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", CREC_PEER_FENCES=str(fences), CREC_BIG_INDEX=str(big))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_peer_worker.py")], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=280))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, "\n----\n".join(e[-1500:] for _, e in outs))
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("PEER_RESULT ")]
    assert line, outs[0][0][-2000:]
    res = json.loads(line[0][len("PEER_RESULT "):])
    assert res["exchanges_ok"]
    for sh in (res["sharded"], res["sharded_unfused"]):
        assert sh["loss_ok"] and sh["same_loss_everywhere"] and sh["P_err"] <= 2e-6 and sh["Q_err"] <= 2e-6, sh
    for name in ("dp", "dp_two_shot"):  # all-reduce + Adam in one launch, the same bits as the all-reduce followed by bpr_adam_dense
        for dp in (res[name], res[name]["unfused"]):
            assert dp["replicas_identical"] and dp["loss_ok"] and dp["P_err"] <= 2e-6 and dp["Q_err"] <= 2e-6, (name, dp)
        assert res[name]["same_bits_as_unfused"], name


def test_ranks_as_threads_sharded_epoch():
    """(Two ranks only: a rank's kernel waits inside the GPU for its peers' kernels, so every rank needs a hardware queue of its
    own -- a process has four, shared by all its streams; one process per GPU never meets that limit.)"""
    G = 2
    U, I, B, d = 400, 301, 256, 64
    g = np.random.Generator(np.random.PCG64(3))
    P0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32)
    Q0 = (g.standard_normal((I, d)) * 0.01).astype(np.float32)
    ranges = [parallel.user_range(U, r, G) for r in range(G)]
    streams = []
    for r, (lo, hi) in enumerate(ranges):
        n = 2 * B + [40, 7, 1, 256][r]
        streams.append((g.integers(0, hi - lo, n).astype(np.int32), np.minimum(g.zipf(1.3, n) - 1, I - 1).astype(np.int32), g.integers(0, I, n).astype(np.int32)))
    sessions = [runtime.Session(0) for _ in range(G)]
    parallel.init_peer_local(sessions, window_bytes=8 << 20)
    for s in sessions:
        s.ctx.check(s.lib.crec_comm_set_timeout(s.ctx.h, 4.0))
    result, errors = [None] * G, []
    # a rank's kernel waits inside the GPU for its peer's kernel: the two streams must not share a hardware queue.  Streams of
    # different priorities never do (a queue has one priority); two streams of the same priority may, depending on how many the
    # process has created before
    rank_streams = [torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)]
    # ... and no rank may copy from pageable host memory (its tables, its triples) while the other's kernel already waits for it:
    # such a copy drains the device's other streams first -- here the waiting kernel, which waits for THIS rank's next launch.
    # (One process per GPU has only its own streams to drain.)  So: every rank's buffers first, then the epochs.
    ready = threading.Barrier(G)

    def work(r):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(rank_streams[r]):
                s = sessions[r]
                lo, hi = ranges[r]
                st = parallel.ShardedBprState(s, P0[lo:hi], parallel.shard_items(Q0, r, G), I, 1e-3, 0.01, r, G)
                u, i, j = (s.to_dev(x) for x in streams[r])
                torch.cuda.current_stream().synchronize()
                ready.wait(60)
                loss = st.train_epoch(u, i, j, B)
                torch.cuda.current_stream().synchronize()
                ready.wait(60)  # (the read-back below is such a copy as well)
                loss = loss.cpu().numpy()
                torch.cuda.current_stream().synchronize()
                result[r] = (st.P.cpu().numpy(), st.Q.cpu().numpy(), loss)
        except Exception as e:  # noqa: BLE001
            errors.append((r, e))

    threads = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not any(t.is_alive() for t in threads), "a rank hung"
    if errors and all("never published" in str(e) for _, e in errors):
        # A rank's kernel waits INSIDE the GPU for its peer's kernel.  Two streams of ONE process are not guaranteed to be scheduled
        # side by side (they may share a hardware queue, and the queues of one process are not pre-empted against each other): the
        # waits then run into their time-out and the library reports it -- an error, not a hang and not an address (that is what
        # this branch checks).  One process per GPU, the production layout, and the ranks-as-processes tests above do not depend on it.
        for s in sessions:
            s.close()
        pytest.skip("the two ranks' kernels were not co-scheduled on this GPU: %s" % "; ".join("rank %d: %s" % (r, e) for r, e in errors))
    if errors:
        raise errors[0][1]
    P = np.concatenate([x[0] for x in result])
    Q = parallel.unshard_items([x[1] for x in result], I)
    oP, oQ, oloss, _ = oracle.sharded_steps(P0, Q0, ranges, streams, B)
    np.testing.assert_allclose(result[0][2], oloss, rtol=1e-5)
    for r in range(1, G):
        assert np.array_equal(result[r][2], result[0][2])
    np.testing.assert_allclose(P, oP, rtol=0, atol=2e-6)
    np.testing.assert_allclose(Q, oQ, rtol=0, atol=2e-6)
    for s in sessions:
        s.close()


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_with_two_ranks_rehearsed_on_one_gpu(launcher):
    """`python bench.py --gpus 2` (bench.py starts its ranks) and the driver's own line, `python -m torch.distributed.run --nnodes=1
    --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...` (whose workers are told to use the launcher's
    store: the canaries' group must not inherit that), rehearsed on ONE GPU (CREC_BENCH_ONE_GPU=1: every rank on cuda:0, the control
    plane over gloo, the data over the peer windows): the launcher, the transport's canary, the replicated-Q headline epoch between two
    processes, the S-big row-sharded legs at 6 144 and at 65 536 triples per rank (the global-memory index builds at a world of two)
    and rank 0's JSON line -- the whole N > 1 flow of bench.py minus the links."""
    root = os.path.dirname(HERE)
    env = dict(os.environ, CREC_BENCH_ONE_GPU="1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    args = [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--cpu-steps", "0"]
    if launcher == "torchrun":
        args = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", env.pop("MASTER_PORT")] + args
    t0 = time.perf_counter()
    cp = subprocess.run([sys.executable] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    took = time.perf_counter() - t0
    assert cp.returncode == 0, cp.stderr[-2000:]
    assert took < 90, "the two-rank rehearsal took %.0f s" % took  # (the driver's scaling run has a wall-clock limit per world size)
    lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, cp.stdout[-1000:]
    b = json.loads(lines[0])
    assert b["n_gpus"] == 2 and b["transport"] == "peer" and b["peer_canary"]["rc"] == 0 and b["value"] > 0
    sb = b["synthetic_big"]
    assert "error" not in sb and sb["world"] == 2 and sb["triples_per_s"] > 0
    assert "error" not in sb["larger_batch"] and sb["larger_batch"]["triples_per_s"] > 0
    x = sb["exchange_per_rank_per_step"]
    assert x["bytes_rows_sent"] > 0 and abs(x["bytes_rows_sent"] - x["bytes_grads_sent"]) < 0.02 * x["bytes_rows_sent"]
    assert b["leg_seconds"]["canary"] < 45 and "leg_errors" not in b


def test_bench_with_a_failed_canary_still_prints_its_line():
    """A node whose GPUs cannot map each other's windows (forced: CREC_CANARY_FORCE_FAIL=1): both canary forms fail at once, the
    failure is on stderr and in the line; on a real node the run would go on over RCCL -- on ONE shared GPU RCCL cannot carry two
    ranks, so the headline fails, and rank 0 STILL prints exactly one line that says so, with a non-zero exit code."""
    root = os.path.dirname(HERE)
    env = dict(os.environ, CREC_BENCH_ONE_GPU="1", CREC_CANARY_FORCE_FAIL="1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cp = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--cpu-steps", "0"], env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=200)
    assert cp.returncode != 0
    lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, cp.stdout[-1000:]
    b = json.loads(lines[0])
    assert b["value"] is None and "canary" in b["error"] and b["peer_canary"]["rc"] != 0
    assert "canary" in cp.stderr and "failed" in cp.stderr


def test_bench_over_rccl_at_a_world_of_one():
    """The fall-back transport end to end: CREC_TRANSPORT=rccl with the N > 1 code path forced at a world of one (CREC_FORCE_DP=1) --
    the replicated-Q headline epoch with RCCL's all-reduce in the loop and both S-big row-sharded legs through grouped
    ncclSend / ncclRecv on a one-rank communicator."""
    root = os.path.dirname(HERE)
    env = dict(os.environ, CREC_TRANSPORT="rccl", CREC_FORCE_DP="1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    t0 = time.perf_counter()
    cp = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-steps", "0"], env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    assert cp.returncode == 0, cp.stderr[-2000:]
    lines = [ln for ln in cp.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, cp.stdout[-1000:]
    b = json.loads(lines[0])
    assert b["transport"] == "rccl" and b["value"] > 0 and "leg_errors" not in b, b.get("leg_errors")
    sb = b["synthetic_big"]
    assert "error" not in sb and sb["triples_per_s"] > 0 and "error" not in sb["larger_batch"] and sb["larger_batch"]["triples_per_s"] > 0
    assert time.perf_counter() - t0 < 240
