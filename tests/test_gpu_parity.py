"""Parity of the HIP path (through the C ABI) with the CPU oracle.  Needs an MI355X.

Tolerances: index/id work bit-exact; fp32 losses rtol 1e-5 (north_star); parameters after a
step: |delta| <= 2e-6 absolute on values of order 1e-2..1 where the Adam step itself is 1e-3
(fp32 dot products are summed in a different order on the GPU, and Adam divides by sqrt(v));
top-K ids and scores bit-exact against the oracle's MFMA-ordered fp32 chain."""
import os

import numpy as np
import pytest
import torch

import oracle
from cleverrec_amd import _lib, runtime
from cleverrec_amd.data import RankingData, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sess():
    s = runtime.Session(0)
    yield s
    s.close()


def _toy_data(seed, U, I, lo, hi, test_every=2):
    g = np.random.Generator(np.random.PCG64(seed))
    ui_train = {u: [int(x) for x in g.choice(I, size=int(g.integers(lo, hi + 1)), replace=False)] for u in range(U) if u % 7 != 6}
    ui_test = {u: [int(x) for x in g.choice(I, size=int(g.integers(1, 5)), replace=False)] for u in range(0, U, test_every)}
    return RankingData(U, I, ui_train, ui_test)


def _tables(seed, U, I, d, scale=0.01):
    g = np.random.Generator(np.random.PCG64(seed))
    return (g.standard_normal((U, d)) * scale).astype(np.float32), (g.standard_normal((I, d)) * scale).astype(np.float32)


def _run_both(sess, data, d, B, epochs, mode, neg_ratio=4, seed=1, scale=0.01, lr=1e-3, reg=0.01):
    P0, Q0 = _tables(7, data.user_nums, data.item_nums, d, scale)
    st = runtime.BprState(sess, P0, Q0, lr=lr, reg=reg, adam_mode=mode)
    om = oracle.BprModel(P0, Q0, lr=lr, reg=reg, adam_mode={"tf1": 0, "lazy": 1}[mode])
    sess.ctx.seed(seed)
    rng = oracle.Rng(seed)
    losses, olosses = [], []
    for _ in range(epochs):
        u, i, j = sess.ctx.sample_pairwise(data.train_csr, data.item_nums, neg_ratio)
        nb, ou, oi, oj = oracle.sample_pairwise(rng, data.ui_train, data.item_nums, neg_ratio, B)
        assert np.array_equal(u, ou) and np.array_equal(i, oi) and np.array_equal(j, oj)  # bit-exact index stream
        losses.append(st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(j), B).cpu().numpy())
        olosses.append(oracle.train_model_py(om, nb, ou, oi, oj, B)[1])
    return st, om, np.concatenate(losses), np.concatenate(olosses)


def _assert_v_close(got, ov, name):
    """v accumulates (1 - beta2) g^2: an error dg of a gradient sum moves it by 2 (1 - beta2) |g| dg, which is small against the
    LARGEST v of the table wherever g itself cancels -- every element within 1e-5 (north_star's bound) of itself or of that scale"""
    bad = np.abs(got - ov) > 1e-5 * ov + 1e-5 * ov.max()
    assert not bad.any(), "%s: %d of %d elements beyond 1e-5 (of the element or of the table's largest v); worst %.3g" % (
        name, int(bad.sum()), bad.size, float((np.abs(got - ov) / (ov + ov.max())).max()))


def _assert_state_close(st, om, atol):
    """Weights and m within `atol`; of a table's elements 1e-5 (none below 100 000 elements) may sit further out, up to 10 atol: where a
    row's gradient cancels to ~0 the fp32 order of its additions decides the sign Adam steps by (the C5 test's rule; seen: one element
    of 264 000 at 2.6e-6 under -DCREC_EXACT_MATH, none in the default build)."""
    for name in ("P", "Q", "mP", "mQ"):
        dev, ref = getattr(st, name).cpu().numpy().astype(np.float64), np.asarray(getattr(om, name), dtype=np.float64)
        dlt = np.abs(dev - ref)
        over = dlt > atol
        if over.any() and over.sum() <= int(1e-5 * dlt.size) and dlt.max() <= 10 * atol:
            continue
        np.testing.assert_allclose(dev, ref, rtol=0, atol=atol, err_msg=name)
    for name in ("vP", "vQ"):
        _assert_v_close(getattr(st, name).cpu().numpy(), getattr(om, name), name)


@pytest.mark.parametrize("mode", ["tf1", "lazy"])
@pytest.mark.parametrize("d", [4, 16, 20, 32, 64, 128, 256])
def test_bpr_epochs_match_oracle(sess, mode, d):
    data = _toy_data(1, 150, 90, 2, 25)
    st, om, loss, oloss = _run_both(sess, data, d, B=256, epochs=2, mode=mode)
    assert st.adam.step == len(loss) and len(loss) == len(oloss)
    np.testing.assert_allclose(loss, oloss, rtol=1e-5)
    _assert_state_close(st, om, atol=2e-6)


@pytest.mark.parametrize("B", [1, 63, 64, 1000, 6144, 8192])
def test_bpr_batch_sizes_and_ragged_tail(sess, B):
    data = _toy_data(2, 80, 50, 1, 20)
    st, om, loss, oloss = _run_both(sess, data, 32, B=B, epochs=1, mode="tf1", neg_ratio=3)
    np.testing.assert_allclose(loss, oloss, rtol=1e-5)
    _assert_state_close(st, om, atol=2e-6)


def test_bpr_launch_padding_does_not_change_results(sess):
    """The workgroups in front of the row waves are padded (to the chunk's largest count, or with option wide_round to a
    multiple of this batch's own) only to keep every row on one XCD from step to step: bitwise the same tables."""
    data = synthetic("S-ml100k")
    P0, Q0 = _tables(12, data.user_nums, data.item_nums, 64)
    sess.ctx.seed(5)
    u, i, j = (sess.to_dev(a) for a in sess.ctx.sample_pairwise(data.train_csr, data.item_nums, 4))
    n = 6 * 6144 + 55
    out = []
    try:
        for rnd in (0, 1, 7, 8):
            sess.ctx.set_option("wide_round", rnd)
            st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
            loss = st.train_epoch(u[:n], i[:n], j[:n], 6144)
            out.append([t.cpu().numpy() for t in (loss, st.P, st.Q, st.vP, st.mQ)])
    finally:
        sess.ctx.set_option("wide_round", 0)
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("mode,U,I", [("lazy", 300_001, 70_003), ("tf1", 65_537, 66_000), ("lazy", 17_000_000, 257)])
def test_bpr_sparse_ids_in_large_tables(sess, mode, U, I):
    """Row ids that need three or four 8-bit passes of the index kernel's radix sort (17 to 25 significant bits, with a
    one-bit last pass at 65 537 rows), active rows scattered over the whole id range, against the oracle.  The tables
    are as tall as the ids but narrow (d = 4), so the CPU side stays small."""
    g = np.random.Generator(np.random.PCG64(33))
    users = np.unique(np.concatenate([g.integers(0, U, 600), [0, U - 1, U // 2, 65_535 % U, 65_536 % U]]))
    items = np.unique(np.concatenate([g.integers(0, I, 200), [0, I - 1]]))
    ui_train = {int(u): [int(x) for x in g.choice(items, size=int(g.integers(1, 12)), replace=False)] for u in users}
    data = RankingData(U, I, ui_train, {int(users[0]): [int(items[0])]})
    st, om, loss, oloss = _run_both(sess, data, 4, B=512, epochs=2, mode=mode)
    np.testing.assert_allclose(loss, oloss, rtol=1e-5)
    _assert_state_close(st, om, atol=2e-6)


def test_bpr_heavy_duplicates_and_large_scores(sess):
    # 3 users x 6 items: every row is hit hundreds of times per batch; scale 1.0 drives x far from 0
    # (softplus branches, saturated sigmoid)
    data = RankingData(3, 6, {0: [0, 1], 1: [2, 3, 4], 2: [5]}, {0: [2]})
    st, om, loss, oloss = _run_both(sess, data, 16, B=512, epochs=3, mode="tf1", neg_ratio=3, scale=1.5, lr=0.05)
    np.testing.assert_allclose(loss, oloss, rtol=1e-5)
    _assert_state_close(st, om, atol=5e-5)


@pytest.mark.parametrize("U,I,B", [(943, 1349, 6144), (60, 40, 2048)])  # ml-100k's shape (wide groups); a tiny catalogue (heavy rows, chunks)
def test_index_by_two_workgroups_per_side_changes_nothing(U, I, B):
    """The tf1 step's per-chunk index is built by two workgroups per (batch, side) that share the phases behind the sort (one the
    heavy rows / wide groups / row segments / counts, the other the occurrence records); `idx_one_wg` restores the single
    workgroup: same tables and losses, bit for bit."""
    g = np.random.Generator(np.random.PCG64(U + B))
    n = 7 * B + 321
    u = g.integers(0, U, n).astype(np.int32)
    i = np.minimum(g.zipf(1.3, n) - 1, I - 1).astype(np.int32)
    j = g.integers(0, I, n).astype(np.int32)
    P0, Q0 = _tables(11, U, I, 32)
    out = []
    for one in (0, 1):
        s = runtime.Session(0)
        s.ctx.set_option("idx_one_wg", one)
        st = runtime.BprState(s, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
        loss = st.train_epoch(s.to_dev(u), s.to_dev(i), s.to_dev(j), B).cpu().numpy()
        out.append([loss] + [getattr(st, k).cpu().numpy() for k in ("P", "Q", "mP", "mQ", "vP", "vQ")])
        s.close()
    for x, y in zip(*out):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("threshold", [1, 8, 24])
def test_bpr_wide_groups_match_plain_row_waves(sess, threshold):
    """tf1 step with overloaded row waves handed to whole workgroups (option wide = occurrences per row wave above which
    that happens; 1: nearly every group that has records) against the same epoch with the mechanism off: same tables up
    to the order of the fp32 partial sums, every row written exactly once (slots of untouched rows decay identically)."""
    data = synthetic("S-ml100k")
    P0, Q0 = _tables(6, data.user_nums, data.item_nums, 64)
    sess.ctx.seed(3)
    u, i, j = (sess.to_dev(a) for a in sess.ctx.sample_pairwise(data.train_csr, data.item_nums, 4))
    n = 5 * 6144 + 777
    out = []
    try:
        for wide in (0, threshold):
            sess.ctx.set_option("wide", wide)
            st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
            loss = st.train_epoch(u[:n], i[:n], j[:n], 6144)
            out.append([t.cpu().numpy() for t in (loss, st.P, st.Q, st.mP, st.vP, st.mQ, st.vQ)])
        np.testing.assert_allclose(out[1][0], out[0][0], rtol=1e-6)
        for a, b in zip(out[0][1:], out[1][1:]):
            np.testing.assert_allclose(b, a, rtol=0, atol=2e-6)
        # a second run with the mechanism on is bitwise the first: the partial sums have a fixed order
        st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
        st.train_epoch(u[:n], i[:n], j[:n], 6144)
        assert np.array_equal(st.Q.cpu().numpy(), out[1][2]) and np.array_equal(st.vP.cpu().numpy(), out[1][4])
    finally:
        sess.ctx.set_option("wide", -1)


def test_bpr_ml100k_shape_full_epoch(sess):
    """BASELINE config C1 shape (943 x 1349, d=32, B=6144, neg_ratio 4) on synthetic data."""
    data = synthetic("S-ml100k")
    st, om, loss, oloss = _run_both(sess, data, 32, B=6144, epochs=1, mode="tf1")
    assert len(loss) == -(-data.n_train * 4 // 6144)
    np.testing.assert_allclose(loss, oloss, rtol=1e-5)
    _assert_state_close(st, om, atol=2e-6)
    # train_model's return value: mean of the per-batch sums (RankingRecommender.py:61)
    assert float(loss.astype(np.float64).mean()) == pytest.approx(float(oloss.mean()), rel=1e-6)


def test_bpr_is_deterministic_at_full_size(sess):
    """ml-1m shape (C2): two runs from the same state are bitwise identical (no atomics), and the
    tf1 / lazy modes agree on every row the epoch touches in every batch-free sense: lazy leaves
    untouched rows' slots at zero while tf1 keeps them at zero too when m == 0."""
    data = synthetic("S-ml1m")
    P0, Q0 = _tables(3, data.user_nums, data.item_nums, 64)
    sess.ctx.seed(1)
    u, i, j = (sess.to_dev(a) for a in sess.ctx.sample_pairwise(data.train_csr, data.item_nums, 4))
    n = 40 * 6144
    outs = []
    for _ in range(2):
        st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
        loss = st.train_epoch(u[:n], i[:n], j[:n], 6144)
        outs.append((loss.cpu().numpy(), st.P.cpu().numpy(), st.Q.cpu().numpy(), st.vQ.cpu().numpy()))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    # loss of an untrained model ~ B*ln2 + reg terms, and it must go down
    assert outs[0][0][0] == pytest.approx(6144 * np.log(2), rel=1e-3)
    assert outs[0][0][-1] < outs[0][0][0]
    # first 3 batches against the oracle at full size
    om = oracle.BprModel(P0, Q0, lr=1e-3, reg=0.01, adam_mode=0)
    un, inn, jn = (t[:3 * 6144].cpu().numpy() for t in (u, i, j))
    _, ol = oracle.train_model_py(om, 3, un, inn, jn, 6144)
    np.testing.assert_allclose(outs[0][0][:3], ol, rtol=1e-5)


def test_bpr_c2_full_epoch_against_the_oracles(sess):
    """BASELINE configs[1] as one whole epoch: S-ml1m, d = 64, batch 6144, tf1 Adam, every batch of the sampled epoch (455 full
    ones and the ragged last) -- the per-batch losses against the C oracle (RankingRecommender.py:39-47 + BPR.py:31-44 +
    tools.py:83-84 restated in oracle/crec_oracle.c) to 1e-5, and the tables and both Adam slots after the LAST step against the
    torch restatement under the rounding envelope it propagates through all 456 steps (tests/_adam_close.py; the flat bounds of
    the short tests are for tens of steps: fp32 rounding noise of 1e-9 per step grows over an epoch, tools/drift.py)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _adam_close import assert_adam_close
    from oracle import models_torch as MT
    data = synthetic("S-ml1m")
    B = 6144
    P0, Q0 = _tables(3, data.user_nums, data.item_nums, 64)
    sess.ctx.seed(1)
    u, i, j = sess.ctx.sample_pairwise(data.train_csr, data.item_nums, 4)
    nb = -(-len(u) // B)
    assert nb >= 450 and len(u) % B != 0
    st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
    loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(j), B).cpu().numpy()
    assert len(loss) == nb and st.adam.step == nb
    om = oracle.BprModel(P0, Q0, lr=1e-3, reg=0.01, adam_mode=0)
    _, oloss = oracle.train_model_py(om, nb, u, i, j, B)
    np.testing.assert_allclose(loss, oloss, rtol=1e-5)
    mt = MT.BPR(P0, Q0, lr=1e-3, reg=0.01)
    tl = [mt.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], j[b * B:(b + 1) * B]) for b in range(nb)]
    np.testing.assert_allclose(loss, tl, rtol=1e-5)
    report = []
    assert_adam_close(mt, "P", st.P, st.mP, st.vP, report=report)
    assert_adam_close(mt, "Q", st.Q, st.mQ, st.vQ, report=report)
    # ... and the C oracle's tables sit inside the same envelope (two CPU restatements, one GPU kernel: all three agree)
    assert_adam_close(mt, "P", om.P, om.mP, om.vP)
    assert_adam_close(mt, "Q", om.Q, om.mQ, om.vQ)


@pytest.mark.parametrize("mode", ["lazy", "tf1", "deferred"])
def test_bpr_c5_full_size_step_against_the_oracle_on_the_touched_rows(sess, mode):
    """BASELINE configs[4] at its full size -- 10 M users x 1 M items, d = 128, B = 6144, tables and slots 16.9 GB in HBM
    (made on the device; the host never holds them).  Only rows a batch has named ever move (lazy: by construction; TF1's
    dense sweep: a row with m = 0 gets lr_t * 0 / (0 + eps)), so the oracle runs the same batches on the COMPACT problem (the
    touched rows, ids renumbered in order) and every touched row, its m and its v must agree; every other row of all six
    arrays must be bit-identical to what it was.  deferred = TF1's numbers with the lazy step's traffic, after flush()."""
    free = torch.cuda.mem_get_info()[0]
    if free < 40 << 30:
        pytest.skip("needs 40 GB of free HBM, %.0f GB free" % (free / 2 ** 30))
    U, I, d, B, nb = 10_000_000, 1_000_000, 128, 6144, 4
    dev = sess.device
    g = torch.Generator(device=dev).manual_seed(5)
    st = runtime.BprState.__new__(runtime.BprState)
    st.sess = sess
    st.P = torch.empty((U, d), dtype=torch.float32, device=dev).normal_(0, 0.01, generator=g)
    st.Q = torch.empty((I, d), dtype=torch.float32, device=dev).normal_(0, 0.01, generator=g)
    st.mP, st.vP, st.mQ, st.vQ = (torch.zeros_like(t) for t in (st.P, st.P, st.Q, st.Q))
    st.tables = _lib.BprTables(*(t.data_ptr() for t in (st.P, st.Q, st.mP, st.vP, st.mQ, st.vQ)), U, I, d)
    st.adam = _lib.Adam(1e-3, 0.9, 0.999, 1e-8, 0.01, {"lazy": _lib.ADAM_LAZY, "tf1": _lib.ADAM_TF1, "deferred": _lib.ADAM_DEFERRED}[mode],
                        0.9, 0.999, 0)
    st.deferred = None
    if mode == "deferred":
        st.lastP, st.lastQ = torch.zeros(U, dtype=torch.int32, device=dev), torch.zeros(I, dtype=torch.int32, device=dev)
        st.deferred = _lib.BprDeferred(st.lastP.data_ptr(), st.lastQ.data_ptr())
    n = nb * B - 100  # ragged last batch
    u = torch.randint(0, U, (n,), generator=g, device=dev).to(torch.int32)
    w = 1.0 / torch.arange(1, I + 1, dtype=torch.float64, device=dev)  # positives Zipf(1): item 0 ~430 times a batch
    i = torch.searchsorted(torch.cumsum(w / w.sum(), 0), torch.rand(n, generator=g, device=dev, dtype=torch.float64)).clamp_(max=I - 1).to(torch.int32)
    j = torch.randint(0, I, (n,), generator=g, device=dev).to(torch.int32)
    u[:7] = torch.tensor([0, U - 1, U - 1, 0, 1 << 23, (1 << 23) - 1, U - 1], dtype=torch.int32)  # the ends of the id range
    j[:3] = torch.tensor([I - 1, 0, I - 1], dtype=torch.int32)
    uu, qq = torch.unique(u.long()), torch.unique(torch.cat([i, j]).long())
    P0c, Q0c = st.P[uu].cpu().numpy(), st.Q[qq].cpu().numpy()
    P_before, Q_before = st.P.clone(), st.Q.clone()
    loss = st.train_epoch(u, i, j, B).cpu().numpy()
    st.flush()
    # ---- the oracle on the compact problem ----
    om = oracle.BprModel(P0c, Q0c, lr=1e-3, reg=0.01, adam_mode=1 if mode == "lazy" else 0)
    cu, ci, cj = (torch.searchsorted(ids, x.long()).to(torch.int32).cpu().numpy() for ids, x in ((uu, u), (qq, i), (qq, j)))
    _, oloss = oracle.train_model_py(om, nb, cu, ci, cj, B)
    np.testing.assert_allclose(loss, oloss, rtol=1e-5)
    for name, ids in (("P", uu), ("Q", qq)):
        got, gm, gv = (getattr(st, pre + name)[ids].cpu().numpy() for pre in ("", "m", "v"))
        # m is linear in the gradient: tight.  A weight whose gradient cancels to ~1e-8 (|m| ~ 1e-9, inside Adam's eps) turns the
        # order of the fp32 additions into a fraction of a step: all but 1e-5 of the elements within 2e-6, none beyond 2 % of a step
        np.testing.assert_allclose(gm, getattr(om, "m" + name), rtol=1e-5, atol=1e-7, err_msg="m" + name)
        # v accumulates (1 - beta2) g^2: an error dg of the gradient sum moves it by 2 (1 - beta2) |g| dg, small against the
        # LARGEST v of the table wherever g itself cancels -- within 1e-5 of the element or of that scale
        ov = getattr(om, "v" + name)
        assert (np.abs(gv - ov) <= 1e-5 * ov + 1e-5 * ov.max()).all(), "v" + name
        dw = np.abs(got - getattr(om, name))
        assert dw.max() <= 2e-5 and (dw > 2e-6).mean() <= 1e-5, "%s: max %.3g, %.3g of the elements beyond 2e-6" % (name, dw.max(), (dw > 2e-6).mean())
    # ---- nothing else moved: rows outside the batches keep their bits, their slots stay zero ----
    for name, ids, before in (("P", uu, P_before), ("Q", qq, Q_before)):
        touched = torch.zeros(before.shape[0], dtype=torch.bool, device=dev)
        touched[ids] = True
        moved = (getattr(st, name) != before).any(dim=1)
        assert not bool((moved & ~touched).any()), name + ": a row outside the batches changed"
        assert bool(moved[ids].all()), name + ": a touched row kept its value"
        del moved
        for pre in ("m", "v"):
            assert not bool(((getattr(st, pre + name) != 0).any(dim=1) & ~touched).any()), pre + name + ": a slot row outside the batches is not zero"


def test_bpr_argument_checks(sess):
    P0, Q0 = _tables(0, 10, 10, 260)  # rows wider than 64 lanes x float4 (a width that is no multiple of 4 is padded by the state)
    st = runtime.BprState(sess, P0, Q0, 1e-3, 0.01)
    z = sess.zeros((8,), torch.int32)
    with pytest.raises(_lib.CrecError, match="embed_size"):
        st.train_epoch(z, z, z, 4)
    P0, Q0 = _tables(0, 10, 10, 8)
    st = runtime.BprState(sess, P0, Q0, 1e-3, 0.01)
    with pytest.raises(_lib.CrecError, match="batch_size"):
        st.train_epoch(z, z, z, 1 << 20)  # (tf1 beyond 8192 runs as deferred Adam + flush, tests/test_gpu_bigbatch.py; the limit is 524 288)
    loss = st.train_epoch(z[:0], z[:0], z[:0], 4)  # empty epoch: nothing happens
    assert loss.numel() == 0 and st.adam.step == 0


# ---------------------------------------------------------------------------------------------
# evaluation
# ---------------------------------------------------------------------------------------------
def _eval_both(sess, data, P, Q, K, ascending=False, users=None):
    users = data.test_users if users is None else np.asarray(users, dtype=np.int32)
    seen_items = data.seen_csr[1] if len(data.seen_csr[1]) else np.zeros(1, np.int32)
    ids, sc = runtime.eval_topk(sess, sess.to_dev(P), sess.to_dev(Q), sess.to_dev(users), sess.to_dev(data.seen_csr[0]),
                                sess.to_dev(seen_items), K, ascending)
    osc = oracle.scores(P, Q, users, order=1)
    oids, oval = oracle.topk(osc, users, data.seen_csr[0], data.seen_csr[1], K, ascending)
    return ids, sc, oids, oval, users


@pytest.mark.parametrize("d", [4, 32, 64, 128, 256])
@pytest.mark.parametrize("ascending", [False, True])
@pytest.mark.parametrize("shape", ["4", "8", "8-search"])  # waves per workgroup of eval_sweep (8: where K and d allow it);
def test_eval_topk_bit_exact(sess, d, ascending, shape):  # "-search": binary-search seen filter, no bitmaps
    data = _toy_data(5, 333, 700, 1, 120, test_every=1)
    P, Q = _tables(11, data.user_nums, data.item_nums, d, scale=1.0)
    Q[10] = Q[9]
    Q[500] = Q[9]  # exact score ties across tiles -> lower id first
    try:
        sess.ctx.set_option("eval_no_bitmap", int(shape.endswith("-search")))
        sess.ctx.set_option("eval_shape", int(shape.split("-")[0]))
        ids, sc, oids, oval, _ = _eval_both(sess, data, P, Q, 20, ascending)
    finally:
        sess.ctx.set_option("eval_no_bitmap", 0)
        sess.ctx.set_option("eval_shape", 0)
    assert np.array_equal(ids.cpu().numpy(), oids)
    assert np.array_equal(sc.cpu().numpy(), oval)


def test_eval_topk_edges(sess):
    # fewer than K unseen items -> -1 padding; users absent from ui_train skip the filter;
    # a single user; K = 1 and K = 32
    I = 40
    ui_train = {0: list(range(0, 35)), 1: list(range(I)), 3: [7]}
    data = RankingData(5, I, ui_train, {0: [36], 1: [0], 2: [5], 3: [1], 4: [2]})
    P, Q = _tables(12, 5, I, 16, scale=1.0)
    for K, shape in ((1, "4"), (1, "8"), (20, "4"), (20, "8"), (24, "8"), (32, "8")):  # K = 32 only has the 4-wave shape
        sess.ctx.set_option("eval_shape", int(shape))
        try:
            ids, sc, oids, oval, _ = _eval_both(sess, data, P, Q, K)
        finally:
            sess.ctx.set_option("eval_shape", 0)
        assert np.array_equal(ids.cpu().numpy(), oids) and np.array_equal(sc.cpu().numpy(), oval)
    ids = ids.cpu().numpy()
    assert (ids[0] >= 0).sum() == 5 and (ids[1] == -1).all() and (ids[2] >= 0).all()
    ids1, sc1, oids1, _, _ = _eval_both(sess, data, P, Q, 20, users=[4])
    assert np.array_equal(ids1.cpu().numpy(), oids1)


def test_eval_metrics_match_oracle_and_reference_semantics(sess):
    data = _toy_data(6, 500, 900, 1, 200, test_every=1)
    P, Q = _tables(13, data.user_nums, data.item_nums, 64, scale=1.0)
    ids, sc, oids, _, users = _eval_both(sess, data, P, Q, 20)
    du = sess.to_dev(users)
    hr, mrr, ndcg = runtime.rank_metrics(sess, ids, du, [10, 20], sess.to_dev(data.test_csr[0]), sess.to_dev(data.test_csr[1]))
    ohr, omrr, ondcg = oracle.rank_metrics(oids, users, [10, 20], data.test_csr[0], data.test_csr[1])
    np.testing.assert_allclose(hr.cpu().numpy(), ohr, rtol=0, atol=0)
    np.testing.assert_allclose(mrr.cpu().numpy(), omrr, rtol=1e-14)
    np.testing.assert_allclose(ndcg.cpu().numpy(), ondcg, rtol=1e-12)
    # and against the statement-for-statement Python restatement of test_model_rs
    H, M, N = oracle.test_model_rs_py(lambda cur: oracle.scores(P, Q, cur, 1), [int(x) for x in users], data.ui_train,
                                      data.ui_test, [10, 20], 128)
    np.testing.assert_allclose(ndcg.cpu().numpy()[0], N[0], rtol=1e-12)
    np.testing.assert_allclose(hr.cpu().numpy()[1], H[1], rtol=0)


def test_eval_full_size_properties(sess):
    """ml-1m shape (C2): size-independent properties of the fused GEMM + top-K at full size."""
    data = synthetic("S-ml1m")
    P, Q = _tables(14, data.user_nums, data.item_nums, 64, scale=0.3)
    K = 20
    ids, sc = runtime.eval_topk(sess, sess.to_dev(P), sess.to_dev(Q), sess.to_dev(data.test_users),
                                sess.to_dev(data.seen_csr[0]), sess.to_dev(data.seen_csr[1]), K)
    ids, sc = ids.cpu().numpy(), sc.cpu().numpy()
    assert ids.shape == (len(data.test_users), K) and (ids >= 0).all() and (ids < data.item_nums).all()
    assert (np.diff(sc, axis=1) <= 0).all()  # best first
    assert all(len(set(r)) == K for r in ids[::97])  # no repeats
    ptr, items = data.seen_csr
    for r in range(0, len(ids), 53):  # never a training item
        uu = data.test_users[r]
        assert not set(ids[r]) & set(items[ptr[uu]:ptr[uu + 1]])
    # returned scores are the dot products, and nothing outside the list beats its last entry
    full = P[data.test_users[::211]].astype(np.float64) @ Q.T.astype(np.float64)
    for n, r in enumerate(range(0, len(ids), 211)):
        np.testing.assert_allclose(sc[r], full[n][ids[r]], rtol=1e-5, atol=1e-6)
        uu = data.test_users[r]
        mask = np.ones(data.item_nums, bool)
        mask[items[ptr[uu]:ptr[uu + 1]]] = False
        mask[ids[r]] = False
        assert full[n][mask].max() <= sc[r][-1] + 1e-5
    # a spot check of whole rows against the oracle, bit for bit
    sub = data.test_users[:256]
    oids, oval = oracle.topk(oracle.scores(P, Q, sub, 1), sub, ptr, items, K)
    assert np.array_equal(ids[:256], oids) and np.array_equal(sc[:256], oval)


def test_model_class_end_to_end(sess, tmp_path):
    """The reference-shaped surface: BPR(sess, data, configs, logger).run_model() on a small
    dataset, against an oracle run of the same protocol (sampler stream from np.random)."""
    import logging
    from cleverrec_amd.model.ranking.BPR import BPR
    data = _toy_data(8, 200, 120, 3, 30)
    configs = {"recommender": "BPR", "epoches": "3", "batch_size": "512", "test.batch_size": "64", "lr": "0.001",
               "test.neg_samples": "0", "is_pairwise": "True", "init_method": "normal", "stddev": "0.01",
               "loss_func": "bpr", "optimizer": "Adam", "saved_dir": str(tmp_path), "test.interval": "1",
               "topk": "[10,20]", "neg_ratio": "4", "embed_size": "32", "reg": "0.01", "data.split_way": "rs"}
    m = BPR(sess, data, configs, logging.getLogger("t"))
    P0, Q0 = _tables(9, data.user_nums, data.item_nums, 32)
    m.build_model(P0, Q0)
    om = oracle.BprModel(P0, Q0, lr=1e-3, reg=0.01)
    np.random.seed(4)
    rng = oracle.Rng(4)
    for _ in range(2):
        avg = m.train_model()
        nb, ou, oi, oj = oracle.sample_pairwise(rng, data.ui_train, data.item_nums, 4, 512)
        oavg, _ = oracle.train_model_py(om, nb, ou, oi, oj, 512)
        assert avg == pytest.approx(oavg, rel=1e-5)
    HR, MRR, NDCG = m.test_model_rs()
    Pn, Qn = m.P.cpu().numpy(), m.Q.cpu().numpy()
    H, M, N = oracle.test_model_rs_py(lambda cur: oracle.scores(Pn, Qn, cur, 1), m.test_users, data.ui_train, data.ui_test,
                                      [10, 20], 64)
    for kid in range(2):
        np.testing.assert_allclose(HR[kid], H[kid], rtol=0)
        np.testing.assert_allclose(NDCG[kid], N[kid], rtol=1e-12)
        np.testing.assert_allclose(MRR[kid], M[kid], rtol=1e-12)
    assert os.path.exists(m.save())


def test_grid_search_driver(sess, tmp_path):
    """main_tuning.run_grid: one model per grid point, each the same as building that model alone (same np.random
    state -> same sampler stream and the same initial tables -> the same best epoch and metrics)."""
    import logging
    import main_tuning
    from cleverrec_amd.model.ranking.BPR import BPR
    data = _toy_data(31, 120, 80, 3, 20)
    configs = {"recommender": "BPR", "model_type": "ranking", "epoches": "2", "batch_size": "256", "test.batch_size": "64",
               "lr": "0.001", "test.neg_samples": "0", "is_pairwise": "True", "init_method": "normal", "stddev": "0.01",
               "loss_func": "bpr", "optimizer": "Adam", "saved_dir": str(tmp_path), "test.interval": "1", "topk": "[10,20]",
               "neg_ratio": "[2,4]", "embed_size": "[8,32]", "reg": "0.01", "data.split_way": "rs"}
    import torch
    np.random.seed(5)
    torch.manual_seed(5)
    results = main_tuning.run_grid(sess, data, configs, logging.getLogger("t"))
    assert [r[0] for r in results] == [{"embed_size": 8, "reg": 0.01, "neg_ratio": 2}, {"embed_size": 8, "reg": 0.01, "neg_ratio": 4},
                                       {"embed_size": 32, "reg": 0.01, "neg_ratio": 2}, {"embed_size": 32, "reg": 0.01, "neg_ratio": 4}]
    np.random.seed(5)
    torch.manual_seed(5)
    alone = BPR(sess, data, dict(configs, embed_size=8, reg=0.01, neg_ratio=2), logging.getLogger("t")).run_model()
    assert alone[0] == results[0][1][0] and alone[1] == results[0][1][1]


def test_data_parallel_epoch_world1_matches_oracle(sess):
    """crec_bpr_train_epoch_dp with a 1-rank RCCL communicator: the same kernels, collectives and dense
    Adam the multi-GPU bench runs, against the oracle's data-parallel restatement and the fused step."""
    from cleverrec_amd.parallel import DpBprState
    data = _toy_data(21, 260, 140, 2, 30)
    d, B = 64, 512
    P0, Q0 = _tables(22, data.user_nums, data.item_nums, d)
    s2 = runtime.Session(0)  # its own ctx: a communicator is bound to a ctx for life
    try:
        st = DpBprState(s2, P0, Q0, lr=1e-3, reg=0.01, rank=0, world=1)
        ref = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
        om = oracle.BprModel(P0, Q0, lr=1e-3, reg=0.01)
        s2.ctx.seed(3)
        u, i, j = s2.ctx.sample_pairwise(data.train_csr, data.item_nums, 4)
        loss = st.train_epoch(s2.to_dev(u), s2.to_dev(i), s2.to_dev(j), B).cpu().numpy()
        rloss = ref.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(j), B).cpu().numpy()
        ol = []
        for b in range(len(loss)):
            sl = slice(b * B, (b + 1) * B)
            lr_t = float(om.sched.next(1)[0])
            dQ, l = oracle.bpr_dp_local(om, u[sl], i[sl], j[sl], lr_t)
            oracle.adam_dense(om, dQ, lr_t)
            ol.append(l)
        np.testing.assert_allclose(loss, ol, rtol=1e-5)
        np.testing.assert_allclose(loss, rloss, rtol=1e-6)
        for name in ("P", "Q", "mQ"):
            np.testing.assert_allclose(getattr(st, name).cpu().numpy(), getattr(om, name), rtol=0, atol=2e-6, err_msg=name)
            np.testing.assert_allclose(getattr(st, name).cpu().numpy(), getattr(ref, name).cpu().numpy(), rtol=0, atol=2e-6)
    finally:
        s2.close()


@pytest.mark.parametrize("neg_samples", [0, 30])
def test_model_loo_matches_reference_semantics(sess, tmp_path, neg_samples):
    """test_model_loo: only ui_test[u] (sampled negatives + held-out items) competes (RankingRecommender.py:250-299)."""
    import logging
    from cleverrec_amd.model.ranking.BPR import BPR
    g = np.random.Generator(np.random.PCG64(31))
    U, I, d = 90, 400, 32
    ui_train = {u: [int(x) for x in g.choice(I, size=int(g.integers(3, 20)), replace=False)] for u in range(U)}
    ui_test = {}
    for u in range(0, U, 2):  # negatives first, then the held-out item(s), like RankingPreprocess._split_data
        rest = np.setdiff1d(np.arange(I), ui_train[u])
        pick = g.choice(rest, size=neg_samples + (1 if neg_samples == 0 else 2), replace=False)
        ui_test[u] = [int(x) for x in pick]
    data = RankingData(U, I, ui_train, ui_test)
    configs = {"recommender": "BPR", "epoches": "1", "batch_size": "512", "test.batch_size": "16", "lr": "0.001",
               "test.neg_samples": str(neg_samples), "is_pairwise": "True", "init_method": "normal", "stddev": "0.01",
               "loss_func": "bpr", "optimizer": "Adam", "saved_dir": str(tmp_path), "test.interval": "1", "topk": "[5,10]",
               "neg_ratio": "4", "embed_size": str(d), "reg": "0.01", "data.split_way": "loo"}
    m = BPR(sess, data, configs, logging.getLogger("t"))
    P0, Q0 = _tables(33, U, I, d, scale=1.0)
    m.build_model(P0, Q0)
    HR, MRR, NDCG = m.test_model_loo()
    H, M, N = oracle.test_model_loo_py(lambda u, items: oracle.scores(P0, Q0[items], [u], 1)[0], m.test_users, data.ui_test,
                                       neg_samples, [5, 10], 16)
    for kid in range(2):
        np.testing.assert_allclose(HR[kid], H[kid], rtol=0)
        np.testing.assert_allclose(MRR[kid], M[kid], rtol=1e-12)
        np.testing.assert_allclose(NDCG[kid], N[kid], rtol=1e-12)


@pytest.mark.parametrize("opt,mode", [("sgd", 2), ("adagrad", 3)])
@pytest.mark.parametrize("d", [16, 64, 128])
def test_bpr_sgd_and_adagrad_match_oracle(sess, opt, mode, d):
    """optimizer = SGD / Adagrad (utils/tools.py:79-87): touched rows only, duplicates summed first; Adagrad's
    accumulator starts at 0.1.  Heavy rows (a 60-item catalogue at B = 512) and a ragged tail included."""
    data = _toy_data(3, 120, 60, 2, 20)
    P0, Q0 = _tables(7, data.user_nums, data.item_nums, d, 0.05)
    lr = 0.05
    st = runtime.BprState(sess, P0, Q0, lr=lr, reg=0.01, adam_mode=opt)
    om = oracle.BprModel(P0, Q0, lr=lr, reg=0.01, adam_mode=mode)
    sess.ctx.seed(4)
    rng = oracle.Rng(4)
    for _ in range(2):
        u, i, j = sess.ctx.sample_pairwise(data.train_csr, data.item_nums, 4)
        nb, ou, oi, oj = oracle.sample_pairwise(rng, data.ui_train, data.item_nums, 4, 512)
        loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(j), 512).cpu().numpy()
        oloss = oracle.train_model_py(om, nb, ou, oi, oj, 512)[1]
        np.testing.assert_allclose(loss, oloss, rtol=1e-5)
    for name in ("P", "Q") + (("mP", "mQ") if opt == "adagrad" else ()):
        np.testing.assert_allclose(getattr(st, name).cpu().numpy(), getattr(om, name), rtol=2e-6, atol=2e-6, err_msg=name)
    if opt == "sgd":  # SGD keeps no slots: they must still be zero
        assert float(st.mP.abs().max()) == 0.0 and float(st.vQ.abs().max()) == 0.0


def test_model_class_with_sgd_and_adagrad(sess, tmp_path):
    """optimizer=SGD / Adagrad in the .properties dict reaches the kernels through the class surface."""
    import logging
    from cleverrec_amd.model.ranking.BPR import BPR
    data = _toy_data(5, 90, 70, 3, 15)
    for name, mode in (("SGD", 2), ("Adagrad", 3)):
        configs = {"recommender": "BPR", "epoches": "1", "batch_size": "256", "test.batch_size": "64", "lr": "0.05",
                   "test.neg_samples": "0", "is_pairwise": "True", "init_method": "normal", "stddev": "0.01", "loss_func": "bpr",
                   "optimizer": name, "saved_dir": str(tmp_path), "test.interval": "1", "topk": "[10,20]", "neg_ratio": "4",
                   "embed_size": "32", "reg": "0.01", "data.split_way": "rs", "sampler_prefetch": "False"}
        m = BPR(sess, data, configs, logging.getLogger("t"))
        P0, Q0 = _tables(2, data.user_nums, data.item_nums, 32)
        m.build_model(P0, Q0)
        om = oracle.BprModel(P0, Q0, lr=0.05, reg=0.01, adam_mode=mode)
        np.random.seed(9)
        rng = oracle.Rng(9)
        avg = m.train_model()
        nb, ou, oi, oj = oracle.sample_pairwise(rng, data.ui_train, data.item_nums, 4, 256)
        oavg, _ = oracle.train_model_py(om, nb, ou, oi, oj, 256)
        assert avg == pytest.approx(oavg, rel=1e-5)
        np.testing.assert_allclose(m.Q.cpu().numpy(), om.Q, rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("d", [32, 128])
def test_bpr_deferred_adam_equals_tf1(sess, d):
    """adam_mode deferred (rows replay the steps they sat out when touched, all rows at flush) against the TF1 sweep of every
    row at every step: the fused tf1 kernel and the oracle, on the S-ml100k epoch (BASELINE config C1 shape)."""
    data = synthetic("S-ml100k")
    P0, Q0 = _tables(3, data.user_nums, data.item_nums, d)
    sess.ctx.seed(2)
    u, i, j = (sess.to_dev(a) for a in sess.ctx.sample_pairwise(data.train_csr, data.item_nums, 4))
    n = 12 * 2048 + 100
    st_d = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="deferred")
    st_t = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
    l_d = torch.cat([st_d.train_epoch(u[:n // 2], i[:n // 2], j[:n // 2], 2048), st_d.train_epoch(u[n // 2:n], i[n // 2:n], j[n // 2:n], 2048)])
    l_t = torch.cat([st_t.train_epoch(u[:n // 2], i[:n // 2], j[:n // 2], 2048), st_t.train_epoch(u[n // 2:n], i[n // 2:n], j[n // 2:n], 2048)])
    np.testing.assert_allclose(l_d.cpu().numpy(), l_t.cpu().numpy(), rtol=1e-5)
    # before the flush the untouched rows lag behind; after it every row is where TF1 has it
    assert not np.allclose(st_d.P.cpu().numpy(), st_t.P.cpu().numpy(), rtol=0, atol=2e-6)
    st_d.flush()
    for name in ("P", "Q", "mP", "mQ"):
        np.testing.assert_allclose(getattr(st_d, name).cpu().numpy(), getattr(st_t, name).cpu().numpy(), rtol=0, atol=2e-6, err_msg=name)
    for name in ("vP", "vQ"):
        _assert_v_close(getattr(st_d, name).cpu().numpy(), getattr(st_t, name).cpu().numpy(), name)
    assert int(st_d.lastP.min()) == int(st_d.lastQ.min()) == st_d.adam.step == st_t.adam.step


def test_bpr_deferred_adam_long_gaps_against_oracle(sess):
    """Rows that sit out more than CREC_REPLAY_MAX = 1024 steps (m has underflowed: the rest of the gap is v *= beta2^n),
    rows touched again after a long gap, and rows never touched, against the oracle's TF1 sweep: 1 500 steps of B = 4."""
    g = np.random.Generator(np.random.PCG64(17))
    U, I, d, B, steps = 60, 50, 8, 4, 1500
    P0, Q0 = _tables(9, U, I, d, 0.1)
    u = g.integers(10, U, steps * B).astype(np.int32)  # users 0..9 only at the very beginning and the very end
    u[:B] = np.arange(B)
    u[-B:] = np.arange(B)
    i = g.integers(5, I, steps * B).astype(np.int32)   # items 0..4 never
    j = g.integers(5, I, steps * B).astype(np.int32)
    i[:B] = 5 + np.arange(B)
    from oracle import models_torch as MT
    from _adam_close import assert_adam_close
    st = runtime.BprState(sess, P0, Q0, lr=1e-2, reg=0.01, adam_mode="deferred")
    om = MT.BPR(P0, Q0, lr=1e-2, reg=0.01)  # the TF1 sweep (every row every step) with the rounding envelope
    loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(j), B).cpu().numpy()
    # 1 500 steps of lr = 1e-2 on batches of four: two correct fp32 trajectories drift apart by the accumulated rounding
    # noise, which the restatement tracks per element (bounds()).  A step's loss may differ by 1e-5 of itself plus what that
    # drift is worth in the loss (first order: sum |dL/dw| * tol_var); the first steps, before any drift, are held to 1e-5 alone.
    for b in range(steps):
        sl = slice(b * B, (b + 1) * B)
        slack = om.loss_slack(u[sl], i[sl], j[sl])
        ol = om.step(u[sl], i[sl], j[sl])
        assert abs(loss[b] - ol) <= 1e-5 * abs(ol) + 2.0 * slack, (b, loss[b], ol, slack)
        if b < 50:
            assert abs(loss[b] - ol) <= 1e-5 * abs(ol), (b, loss[b], ol)
    st.flush()
    drift = max(float(om.bounds(k)[0].max()) for k in ("P", "Q"))  # how far a parameter of either trajectory may have wandered
    for name in ("P", "Q"):
        assert_adam_close(om, name, getattr(st, name), None, None, lr=1e-2)
        # m is a running mean of gradient sums -s (q_i - q_j) + reg p (|s| <= 1): besides the rounding of the sums themselves
        # (tol_m) it inherits the parameters' drift, at most ~2 rows' worth per term; 4 x covers the second-order terms
        mr = om.opt.m[name].numpy().astype(np.float64)
        assert (np.abs(getattr(st, "m" + name).cpu().numpy() - mr) <= om.bounds(name)[1] + 1e-6 * np.abs(mr) + 1e-10 + 4 * drift).all(), "m" + name
    assert drift < 1e-3  # the envelope stays far below the parameters themselves (|P|, |Q| ~ 0.1): it constrains
    np.testing.assert_allclose(st.Q.cpu().numpy()[:5], Q0[:5], rtol=0, atol=0)  # never touched: m = v = 0, nothing moves
    for name in ("vP", "vQ"):  # beyond 1 024 idle steps v is scaled by one powf instead of n multiplications (a few ulp), on top of its envelope
        ov = om.opt.v[name[1:]].numpy().astype(np.float64)
        tol = om.bounds(name[1:])[2] + 1e-5 * ov + 1e-5 * ov.max() + 8 * drift * np.sqrt(ov.max())  # (1 - b2) 2 |g| dg summed over ~1 / (1 - b2) steps
        assert (np.abs(getattr(st, name).cpu().numpy() - ov) <= tol).all(), name


def test_counts_published_by_the_index_kernel_equal_the_copied_ones(sess):
    """tf1 mode: the per-chunk counts that size the step launches reach the host through page-locked memory written by the
    index kernel's last workgroup (csrc/bpr_train.hip, IndexWs::host_counts); `no_zero_copy` restores the copy + stream
    synchronisation.  Same launches either way: bit-identical tables and losses, over several chunks and calls."""
    U, I, d, B = 500, 60, 32, 512  # a small catalogue: heavy rows and wide groups in every batch
    g = np.random.Generator(np.random.PCG64(12))
    P0, Q0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32), (g.standard_normal((I, d)) * 0.01).astype(np.float32)
    n = 150 * B + 77  # more than one chunk of 64 batches, a ragged tail
    u, i, j = (g.integers(0, m, n).astype(np.int32) for m in (U, I, I))
    out = []
    for flag in (0, 1):
        sess.ctx.set_option("no_zero_copy", flag)
        try:
            st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
            du, di, dj = (sess.to_dev(x) for x in (u, i, j))
            losses = [st.train_epoch(du, di, dj, B).cpu().numpy(), st.train_epoch(du[:5 * B], di[:5 * B], dj[:5 * B], B).cpu().numpy()]
            out.append((np.concatenate(losses), st.P.cpu().numpy(), st.Q.cpu().numpy(), st.vQ.cpu().numpy()))
        finally:
            sess.ctx.set_option("no_zero_copy", 0)
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)


def test_csv_to_metrics_pipeline(sess, tmp_path):
    """f2 + a16 on the GPU box: ratings.csv -> RankingPreprocess (filter, re-index, split, CSR; checked against the
    reference's own output on the same file, tests/golden/preprocess_syn.npz) -> BPR(sess, data, configs, logger) -> one
    epoch and the rs evaluation, against the oracle's trainer on the reference's CSR."""
    import logging
    from cleverrec_amd.model.RankingPreprocess import RankingPreprocess
    from cleverrec_amd.model.ranking.BPR import BPR
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess_syn.npz"))
    d = tmp_path / "syn"
    d.mkdir()
    with open(d / "ratings.csv", "w") as f:
        for a, b in zip(g["csv_u"], g["csv_i"]):
            f.write("%d,%d\n" % (a, b))
    cfg = {"data.root_dir": str(tmp_path), "data.dataset": "syn", "data.file_name": "ratings.csv", "data.sep": ",",
           "data.format": "UI", "data.split_way": "rs", "data.split_ratio": "[0.7,0.2,0.1]", "data.split_by_time": "False",
           "data.user_min": "0", "data.item_min": "5", "test.neg_samples": "0", "recommender": "BPR"}
    np.random.seed(int(g["seed"]))
    data = RankingPreprocess(cfg, logging.getLogger("t"))
    for mine, ref in zip(data.train_csr, ("train_keys", "train_ptr", "train_items")):
        assert np.array_equal(mine, g[ref]), ref
    configs = dict(cfg, **{"epoches": "1", "batch_size": "256", "test.batch_size": "64", "lr": "0.001", "is_pairwise": "True",
                           "init_method": "normal", "stddev": "0.01", "loss_func": "bpr", "optimizer": "Adam",
                           "saved_dir": str(tmp_path), "test.interval": "1", "topk": "[5,10]", "neg_ratio": "4",
                           "embed_size": "16", "reg": "0.01"})
    m = BPR(sess, data, configs, logging.getLogger("t"))
    P0, Q0 = _tables(21, data.user_nums, data.item_nums, 16)
    m.build_model(P0, Q0)
    om = oracle.BprModel(P0, Q0, lr=1e-3, reg=0.01)
    ref_train = {int(k): [int(x) for x in g["train_items"][g["train_ptr"][n]:g["train_ptr"][n + 1]]] for n, k in enumerate(g["train_keys"])}
    np.random.seed(4)
    avg = m.train_model()
    nb, ou, oi, oj = oracle.sample_pairwise(oracle.Rng(4), ref_train, int(g["item_nums"]), 4, 256)
    oavg, _ = oracle.train_model_py(om, nb, ou, oi, oj, 256)
    assert avg == pytest.approx(oavg, rel=1e-5)
    HR, MRR, NDCG = m.test_model_rs()
    Pn, Qn = m.P.cpu().numpy(), m.Q.cpu().numpy()
    H, M, N = oracle.test_model_rs_py(lambda cur: oracle.scores(Pn, Qn, cur, 1), m.test_users, data.ui_train, data.ui_test, [5, 10], 64)
    for kid in range(2):
        np.testing.assert_allclose(HR[kid], H[kid], rtol=0)
        np.testing.assert_allclose(NDCG[kid], N[kid], rtol=1e-12)


@pytest.mark.parametrize("init_method,expect", [("normal", "normal"), ("tnormal", "tnormal"), ("xavier_normal", "xavier_normal"),
                                                ("xavier", "xavier")])
def test_initializers_reach_the_device_tables(sess, tmp_path, init_method, expect):
    """a9 on the GPU box: BPR(...).build_model() without injected tables draws P, Q with get_initializer(init_method, stddev)
    (utils/tools.py:51-63) and the rows the step trains are those draws: moments / supports of the DEVICE tables."""
    import logging
    import math
    import torch
    from cleverrec_amd.model.ranking.BPR import BPR
    data = _toy_data(3, 3000, 400, 3, 12)
    s, d = 0.05, 64
    configs = {"recommender": "BPR", "epoches": "1", "batch_size": "512", "test.batch_size": "64", "lr": "0.001",
               "test.neg_samples": "0", "is_pairwise": "True", "init_method": init_method, "stddev": str(s), "loss_func": "bpr",
               "optimizer": "Adam", "saved_dir": str(tmp_path), "test.interval": "1", "topk": "[10]", "neg_ratio": "1",
               "embed_size": str(d), "reg": "0.01", "data.split_way": "rs"}
    torch.manual_seed(11)
    m = BPR(sess, data, configs, logging.getLogger("t"))
    m.build_model()
    x = m.P.cpu().double()
    n = data.user_nums
    assert m.P.is_cuda and tuple(x.shape) == (n, d) and float(m.state.mP.abs().max()) == 0.0
    if expect == "normal":
        assert abs(float(x.mean())) < 5 * s / math.sqrt(n * d) and abs(float(x.std()) / s - 1) < 0.02
    elif expect == "tnormal":
        assert float(x.abs().max()) <= 2 * s and abs(float(x.std()) / (0.8796 * s) - 1) < 0.02
    elif expect == "xavier_normal":
        sd = math.sqrt(1.3 * 2.0 / (n + d))
        assert float(x.abs().max()) <= 2 * sd and abs(float(x.std()) / (0.8796 * sd) - 1) < 0.02
    else:
        lim = math.sqrt(6.0 / (n + d))
        assert float(x.abs().max()) <= lim and abs(float(x.std()) / (lim / math.sqrt(3)) - 1) < 0.02


@pytest.mark.parametrize("K,bitmap", [(50, True), (50, False), (100, True), (128, False), (65, True)])
def test_topk_lists_longer_than_32(sess, K, bitmap):
    """topk beyond 32 (the reference takes any list, Recommender.py:27): ranked 32 at a time, every pass with the ranks above it
    out of the competition -- ids and scores bit-exact against the oracle's full argsort, in both seen-filter paths, and the
    metrics of the long list against the oracle's cal_ranking_metrics."""
    g = np.random.Generator(np.random.PCG64(K))
    U, I, d = 333, 700, 32
    P, Q = _tables(K, U, I, d)
    ui_train = {u: [int(x) for x in g.choice(I, size=int(g.integers(1, 600 if u % 9 == 0 else 60)), replace=False)] for u in range(U)}
    ui_train[0] = list(range(I - 40))  # fewer unseen items than any of the K tried
    ui_test = {u: [int(x) for x in g.choice(I, size=int(g.integers(1, 8)), replace=False)] for u in range(0, U, 2)}
    data = RankingData(U, I, ui_train, ui_test)
    sess.ctx.set_option("eval_no_bitmap", 0 if bitmap else 1)
    try:
        users = sess.to_dev(data.test_users)
        ids, sc = runtime.eval_topk(sess, sess.to_dev(P), sess.to_dev(Q), users, sess.to_dev(data.seen_csr[0]), sess.to_dev(data.seen_csr[1]), K)
    finally:
        sess.ctx.set_option("eval_no_bitmap", 0)
    osc = oracle.scores(P, Q, data.test_users, order=1)
    oids, oval = oracle.topk(osc, data.test_users, data.seen_csr[0], data.seen_csr[1], K)
    assert np.array_equal(ids.cpu().numpy(), oids) and np.array_equal(sc.cpu().numpy(), oval)
    assert (oids[0, 40:] == -1).all() and (oids[0, :40] >= 0).all()  # user 0 has 40 unseen items: the tail of its list is padding
    ks = [10, min(50, K), K]
    hr, mrr, ndcg = runtime.rank_metrics(sess, ids, users, ks, sess.to_dev(data.test_csr[0]), sess.to_dev(data.test_csr[1]))
    ohr, omrr, ondcg = oracle.rank_metrics(oids, data.test_users, ks, data.test_csr[0], data.test_csr[1])
    np.testing.assert_allclose(hr.cpu().numpy(), ohr, rtol=1e-12)
    np.testing.assert_allclose(mrr.cpu().numpy(), omrr, rtol=1e-12)
    np.testing.assert_allclose(ndcg.cpu().numpy(), ondcg, rtol=1e-12)


@pytest.mark.parametrize("mode", ["tf1", "lazy"])
def test_bpr_embed_size_that_is_no_multiple_of_four(sess, mode):
    """embed_size = 30 (the reference takes any width): the state pads the rows with two zero columns, which stay exactly zero;
    the 30 real columns, the losses and the ranking equal the oracle's at d = 30."""
    U, I, d, B = 300, 200, 30, 512
    g = np.random.Generator(np.random.PCG64(30))
    P0, Q0 = _tables(30, U, I, d)
    n = 3 * B + 77
    u, i, j = g.integers(0, U, n).astype(np.int32), g.integers(0, I, n).astype(np.int32), g.integers(0, I, n).astype(np.int32)
    st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode=mode)
    loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(j), B).cpu().numpy()
    om = oracle.BprModel(P0, Q0, lr=1e-3, reg=0.01, adam_mode=0 if mode == "tf1" else 1)
    _, oloss = oracle.train_model_py(om, len(loss), u, i, j, B)
    np.testing.assert_allclose(loss, oloss, rtol=1e-5)
    assert st.P.shape[1] == 32 and not st.P[:, 30:].any() and not st.Q[:, 30:].any() and not st.mP[:, 30:].any() and not st.vQ[:, 30:].any()
    np.testing.assert_allclose(st.P[:, :30].cpu().numpy(), om.P, rtol=0, atol=2e-6)
    np.testing.assert_allclose(st.Q[:, :30].cpu().numpy(), om.Q, rtol=0, atol=2e-6)


@pytest.mark.parametrize("ascending", [False, True])
def test_topk_of_a_score_matrix_wider_than_the_lds_stage(sess, ascending):
    """crec_topk_scores (the tower models' and the distance models' ranking step) on 90 001 items: the item axis is ranked a stretch
    of 40 000 at a time and the stretches' best once more -- ids bit-exact against the oracle's argsort, ties (quantised scores:
    plenty) to the lower item id"""
    g = np.random.Generator(np.random.PCG64(4))
    Bt, I, K = 37, 90001, 20
    sc = np.round(g.standard_normal((Bt, I)) * 50).astype(np.float32) / 50
    ptr = np.zeros(Bt + 1, np.int64)
    rows = [np.sort(g.choice(I, size=int(g.integers(0, 3000)), replace=False)).astype(np.int32) for _ in range(Bt)]
    np.cumsum([len(r) for r in rows], out=ptr[1:])
    items = np.concatenate(rows)
    users = np.arange(Bt, dtype=np.int32)
    ids, val = runtime.topk_scores(sess, sess.to_dev(sc), sess.to_dev(users), sess.to_dev(ptr), sess.to_dev(items), K, ascending=ascending)
    oids, oval = oracle.topk(sc, users, ptr, items, K, ascending=ascending)
    assert np.array_equal(ids.cpu().numpy(), oids) and np.array_equal(val.cpu().numpy(), oval)
