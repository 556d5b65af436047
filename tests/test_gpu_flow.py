"""bpr_steps_flow (tf1 Adam, all steps of a chunk in ONE dataflow launch: a lane group owns a table row for up to 64 steps and
publishes a version of it per step) against
  * the same kernel launched once per step (option flow_steps = 1: every gather reads the tables themselves, no version
    history, no waiting) -- the same arithmetic by construction, so tables, Adam slots and losses must agree BIT FOR BIT;
  * the launch-per-step kernel bpr_step_fused (option flow = 0) and the CPU oracle, within the suite's usual bounds (the
    two kernels assign rows to waves differently, so their fp32 sums differ in order).
Reference: model/RankingRecommender.py:39-47 (one sess.run per batch), model/ranking/BPR.py:31-44, utils/tools.py:83-84."""
import numpy as np
import pytest

import oracle
from cleverrec_amd import runtime
from cleverrec_amd.data import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sess():
    s = runtime.Session(0)
    yield s
    s.close()


def _stream(seed, U, I, n, zipf=1.2):
    g = np.random.Generator(np.random.PCG64(seed))
    u = g.integers(0, U, n).astype(np.int32)
    i = np.minimum(g.zipf(zipf, n) - 1, I - 1).astype(np.int32)
    j = g.integers(0, I, n).astype(np.int32)
    return u, i, j


def _tables(seed, U, I, d, scale=0.01):
    g = np.random.Generator(np.random.PCG64(seed))
    return (g.standard_normal((U, d)) * scale).astype(np.float32), (g.standard_normal((I, d)) * scale).astype(np.float32)


def _run(sess, opts, P0, Q0, u, i, j, B, calls=1, lr=1e-3):
    """(options: flow = 1 unless given) the stream in `calls` consecutive calls of whole batches (the last one takes the ragged tail)"""
    try:
        opts = dict({"flow": 1}, **opts)  # (the dataflow launch is opt-in)
        for k, v in opts.items():
            sess.ctx.set_option(k, v)
        st = runtime.BprState(sess, P0, Q0, lr=lr, reg=0.01, adam_mode="tf1")
        nb = -(-len(u) // B)
        cuts = [round(c * nb / calls) * B for c in range(calls)] + [len(u)]
        losses = []
        du, di, dj = (sess.to_dev(x) for x in (u, i, j))
        for a, b in zip(cuts[:-1], cuts[1:]):
            if b > a:
                losses.append(st.train_epoch(du[a:b], di[a:b], dj[a:b], B).cpu().numpy())
        return [np.concatenate(losses)] + [getattr(st, n).cpu().numpy() for n in ("P", "Q", "mP", "mQ", "vP", "vQ")], st
    finally:
        for k in opts:
            sess.ctx.set_option(k, {"flow_deal": 1, "flow": -1}.get(k, 0))


CASES = [
    # U, I, d, B, batches (+ a ragged tail), zipf
    (6040, 3706, 64, 6144, 20, 1.2),    # the headline shape, one launch
    (6040, 3706, 64, 6144, 150, 1.2),   # three launches: both histories in use, each reset by the launch after it
    (943, 1349, 32, 6144, 9, 1.2),      # ml-100k's shape: rows with a hundred occurrences per batch
    (150, 90, 4, 256, 70, 1.5),         # one lane per row
    (150, 90, 16, 256, 13, 1.5),
    (333, 777, 20, 1000, 11, 1.3),      # d no multiple of the lane group
    (500, 300, 128, 2048, 10, 1.2),     # two rows per wave
    (200, 100, 256, 512, 10, 1.2),      # a wave per row
    (40, 25, 64, 4096, 8, 1.5),         # every row heavy: hundreds of occurrences per batch
    (17, 1, 8, 64, 5, 1.1),             # one item
]


@pytest.mark.parametrize("U,I,d,B,nb,zipf", CASES)
def test_flow_equals_a_launch_per_step_bit_for_bit(sess, U, I, d, B, nb, zipf):
    u, i, j = _stream(U + d, U, I, nb * B + B // 3 + 1, zipf)
    P0, Q0 = _tables(3, U, I, d)
    a, _ = _run(sess, {}, P0, Q0, u, i, j, B)
    b, _ = _run(sess, {"flow_steps": 1}, P0, Q0, u, i, j, B)
    for x, y, name in zip(a, b, ("loss", "P", "Q", "mP", "mQ", "vP", "vQ")):
        assert np.array_equal(x, y), name
    # ... however the batches are cut into calls and launches, and wherever the rows sit (dealt round-robin by id instead of by weight)
    c, _ = _run(sess, {"flow_steps": 7, "flow_deal": 0}, P0, Q0, u, i, j, B, calls=3)
    for x, y, name in zip(a, c, ("loss", "P", "Q", "mP", "mQ", "vP", "vQ")):
        assert np.array_equal(x, y), name


def test_flow_is_taken_for_the_headline_shape_and_counts_its_steps(sess):
    U, I, d, B = 6040, 3706, 64, 6144
    u, i, j = _stream(5, U, I, 70 * B)
    P0, Q0 = _tables(3, U, I, d)
    sess.ctx.profile(True)
    try:
        _run(sess, {}, P0, Q0, u, i, j, B)
        prof = sess.ctx.profile_read()
        units = sess.ctx.profile_units("bpr_steps_flow")
    finally:
        sess.ctx.profile(False)
    assert prof["bpr_step_fused"][1] == 0
    ms, launches = prof["bpr_steps_flow"]
    assert launches == 2 and units == 70 and ms > 0


@pytest.mark.parametrize("U,I,d,B,nb", [(6040, 3706, 64, 6144, 12), (943, 1349, 32, 6144, 6), (300, 200, 128, 1024, 9)])
def test_flow_against_the_launch_per_step_kernel_and_the_oracle(sess, U, I, d, B, nb):
    u, i, j = _stream(11, U, I, nb * B + 77)
    P0, Q0 = _tables(4, U, I, d)
    a, st = _run(sess, {}, P0, Q0, u, i, j, B)
    b, _ = _run(sess, {"flow": 0}, P0, Q0, u, i, j, B)
    om = oracle.BprModel(P0, Q0, lr=1e-3, reg=0.01, adam_mode=0)
    oloss = oracle.train_model_py(om, nb + 1, u, i, j, B)[1]
    np.testing.assert_allclose(a[0], oloss, rtol=1e-5)
    np.testing.assert_allclose(b[0], oloss, rtol=1e-5)
    for x, y, name in zip(a[1:], b[1:], ("P", "Q", "mP", "mQ", "vP", "vQ")):
        if name.startswith("v"):  # within 1e-5 of the element or of the table's largest v (tests/test_gpu_parity.py: _assert_v_close)
            assert (np.abs(x - y) <= 1e-5 * y + 1e-5 * y.max()).all(), name
        else:
            dw = np.abs(x - y)
            assert dw.max() <= 2e-5 and (dw > 2e-6).mean() <= 1e-5, "%s: max %.3g" % (name, dw.max())
    for name, x in zip(("P", "Q", "mP", "mQ"), (a[1], a[2], a[3], a[4])):
        # (a weight whose gradient cancels turns the order of the fp32 additions into a fraction of an Adam step: all but 1e-5 of the
        # elements within 2e-6, none beyond 2 % of a step -- the bound of the C5 test)
        dw = np.abs(x - getattr(om, name))
        assert dw.max() <= 2e-5 and (dw > 2e-6).mean() <= 1e-5, "%s: max %.3g, %.3g of the elements beyond 2e-6" % (name, dw.max(), (dw > 2e-6).mean())
    assert st.adam.step == nb + 1


def test_tables_too_tall_for_resident_waves_take_the_launch_per_step_path(sess):
    U, I, d, B = 200_000, 1000, 16, 1024
    u, i, j = _stream(13, U, I, 5 * B)
    P0, Q0 = _tables(5, U, I, d)
    sess.ctx.profile(True)
    try:
        a, _ = _run(sess, {"tf1_sweep": 1}, P0, Q0, u, i, j, B)  # (77 MB of rows and slots: swept only on request)
        prof = sess.ctx.profile_read()
    finally:
        sess.ctx.profile(False)
    assert prof["bpr_steps_flow"][1] == 0 and prof["bpr_step_fused"][1] >= 4  # (the chunk's first step is launched beside the index kernel: not counted)


@pytest.mark.parametrize("flow", [1, 0])
@pytest.mark.parametrize("U,I,d,B,nb", [(6040, 3706, 64, 6144, 20), (943, 1349, 32, 4096, 90), (300, 200, 16, 1000, 7)])
def test_records_from_lds_ids_are_the_gathered_records(sess, flow, U, I, d, B, nb):
    """tf1 index: with both catalogues below 65 536 rows a batch's three id streams are kept in LDS as 16-bit ids and the occurrence
    records are put together from there (option idx_no_ids16 = 1: gathered from the streams as before).  Same records, so the same
    bits -- for the dataflow launch and the launch per step, few batches per call (two workgroups per side) and many."""
    u, i, j = _stream(3 * U + B, U, I, nb * B + 17)
    P0, Q0 = _tables(U, U, I, d)
    ref, _ = _run(sess, {"flow": flow, "idx_no_ids16": 1}, P0, Q0, u, i, j, B)
    got, _ = _run(sess, {"flow": flow}, P0, Q0, u, i, j, B)
    for x, y, name in zip(ref, got, ("loss", "P", "Q", "mP", "mQ", "vP", "vQ")):
        assert np.array_equal(x, y), name


@pytest.mark.parametrize("flow", [1, 0])
@pytest.mark.parametrize("U,I,d,B,nb", [(6040, 3706, 64, 6144, 20), (6040, 3706, 64, 6144, 67), (333, 777, 20, 1000, 11), (150, 90, 16, 256, 5)])
def test_tables_brought_home_inside_the_loss_reduction_launch_are_the_same_tables(sess, flow, U, I, d, B, nb):
    """A call that ends in the ctx's second buffer set brings its tables home inside the last chunk's loss reduction launch
    (bpr_call_tail); option no_call_tail = 1 keeps the copy a launch of its own.  Same tables, slots and losses, bit for bit, over
    calls of odd and even length, with either step kernel (RankingRecommender.py:39-47: the caller's tables after every call)."""
    u, i, j = _stream(U + nb, U, I, nb * B + B // 5 + 1)
    P0, Q0 = _tables(4, U, I, d)
    a, _ = _run(sess, {"flow": flow}, P0, Q0, u, i, j, B, calls=3)
    b, _ = _run(sess, {"flow": flow, "no_call_tail": 1}, P0, Q0, u, i, j, B, calls=3)
    for x, y, name in zip(a, b, ("loss", "P", "Q", "mP", "mQ", "vP", "vQ")):
        assert np.array_equal(x, y), name
