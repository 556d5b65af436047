"""Batches beyond what a workgroup sorts in LDS (batch_size > 8192; the reference takes any batch_size, model/Recommender.py:17).

  * the index built in global memory (csrc/index_big.hip) equals the in-LDS index (bpr_index_batches) BIT FOR BIT on batches both
    can build -- descriptors, processing order, front-region sizes;
  * the BPR step on batches of 16 384 / 20 000 / 40 000 triples against the oracle (lazy and SGD / Adagrad: touched rows only; deferred
    against the oracle's TF1 sweep), fp32 losses rtol 1e-5, parameters atol 2e-6 as in test_gpu_parity.py;
  * the `big_index` option on a batch of 6144: the same tables and losses, bit for bit, as the default path.
Needs an MI355X."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from cleverrec_amd import runtime

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sess():
    s = runtime.Session(0)
    yield s
    s.close()


def _stream(seed, n, U, I, zipf=None, item_cap=None):
    g = np.random.Generator(np.random.PCG64(seed))
    u = g.integers(0, U, n).astype(np.int32)
    if zipf:
        i = np.minimum(g.zipf(zipf, n) - 1, I - 1).astype(np.int32)
    else:
        i = g.integers(0, item_cap or I, n).astype(np.int32)
    j = g.integers(0, item_cap or I, n).astype(np.int32)
    j = np.where(j == i, (j + 1) % I, j).astype(np.int32)  # (a sampled negative is never the positive)
    return u, i, j


def _index(sess, u, i, j, B, U, I, d, big):
    n = len(u)
    nb = -(-n // B)
    dev = lambda shape: torch.empty(shape, dtype=torch.int32, device=sess.device)  # noqa: E731
    ud, idd, jd, pu, ns = dev((nb, B, 4)), dev((nb, B, 4)), dev((nb, B, 4)), dev((nb, B, 2)), dev((nb, 2))
    a = [sess.to_dev(x) for x in (u, i, j)]
    sess.ctx.check(sess.lib.crec_debug_lazy_index(sess.ctx.h, sess.ptr(a[0]), sess.ptr(a[1]), sess.ptr(a[2]), n, B, U, I, d, int(big),
                                                  sess.ptr(ud), sess.ptr(idd), sess.ptr(jd), sess.ptr(pu), sess.ptr(ns), sess.stream_ptr()))
    return [t.cpu().numpy() for t in (ud, idd, jd, pu, ns)]


@pytest.mark.parametrize("n,B,U,I,d,kw", [
    (6144 * 3 + 100, 6144, 100000, 1000000, 128, dict(zipf=1.2)),   # S-big's shape: a Zipf head of popular positives
    (8192 * 2, 8192, 500, 40, 64, dict()),                          # every row heavy
    (8192 + 1, 8192, 3, 2, 32, dict()),                             # two items, three users: runs of thousands
    (250, 100, 50, 30, 16, dict()),                                 # a ragged last batch
    (7, 1, 5, 5, 4, dict()),                                        # batches of one; d = 4: 64 slots per workgroup
    (4000, 2000, 6040, 3706, 256, dict(zipf=1.5)),                  # d = 256: 16 slots per workgroup
    (6144, 6144, 10_000_000, 1_000_000, 128, dict()),               # 24-bit user ids: three radix digits
])
def test_big_index_equals_lds_index(sess, n, B, U, I, d, kw):
    u, i, j = _stream(n + B, n, U, I, **kw)
    a = _index(sess, u, i, j, B, U, I, d, big=False)
    b = _index(sess, u, i, j, B, U, I, d, big=True)
    for name, x, y in zip(("u_desc", "i_desc", "j_desc", "slot_pu", "nslots"), a, b):
        assert np.array_equal(x, y), "%s differs at %s" % (name, np.argwhere(x != y)[:4].tolist())
    # and it is an index: every batch position appears once in the processing order
    nb = -(-n // B)
    for bt in range(nb):
        m = min(B, n - bt * B)
        assert np.array_equal(np.sort(a[3][bt, :m, 0]), np.arange(m))
        assert a[4][bt, 0] == m


@pytest.mark.parametrize("big", [False, True])
@pytest.mark.parametrize("n,B,U,I,d", [(3 * 1000 + 77, 1000, 5000, 800, 64),   # a Zipf head: front rows across workgroup boundaries
                                       (2 * 2048, 2048, 40, 12, 128),         # every row heavy
                                       (150, 64, 30, 20, 4),                   # d = 4: 64 slots per workgroup
                                       (700, 700, 100000, 70000, 256)])        # nearly every row single
def test_device_index_equals_the_cpu_restatement(sess, n, B, U, I, d, big):
    """Both device builders of the lazy step's index (in-LDS sort, global-memory sort) against oracle/lazy_index.py -- the numpy
    restatement of TF1's unique + segment-sum bookkeeping in the packed form of csrc/lazy_desc.h -- bit for bit: every descriptor,
    the processing order, the front-region size."""
    from oracle import lazy_index
    u, i, j = _stream(n + d, n, U, I, zipf=1.25)
    got = _index(sess, u, i, j, B, U, I, d, big=big)
    spw = 64 if d <= 4 else 16
    for b in range(-(-n // B)):
        sl = slice(b * B, min(n, (b + 1) * B))
        ud, idd, jd, pu, (nb, nf) = lazy_index.batch_index(u[sl], i[sl], j[sl], spw)
        assert np.array_equal(got[0][b, :nb], ud), "u_desc of batch %d" % b
        assert np.array_equal(got[1][b, :nb], idd), "i_desc of batch %d" % b
        assert np.array_equal(got[2][b, :nb], jd), "j_desc of batch %d" % b
        assert np.array_equal(got[3][b, :nb], pu), "slot_pu of batch %d" % b
        assert got[4][b].tolist() == [nb, nf]


def _oracle_steps(P0, Q0, u, i, j, B, lr, reg, mode):
    om = oracle.BprModel(P0, Q0, lr=lr, reg=reg, adam_mode=mode)
    nb = -(-len(u) // B)
    _, ol = oracle.train_model_py(om, nb, u, i, j, B)
    return om, ol


@pytest.mark.parametrize("B,mode", [(16384, "lazy"), (20000, "lazy"), (40000, "lazy"), (16384, "deferred")])
def test_large_batches_against_the_oracle(sess, B, mode):
    U, I, d, lr, reg = 5000, 3000, 32, 1e-3, 0.01
    n = 2 * B + 777
    u, i, j = _stream(B, n, U, I, zipf=1.3)
    g = np.random.Generator(np.random.PCG64(5))
    P0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32)
    Q0 = (g.standard_normal((I, d)) * 0.01).astype(np.float32)
    st = runtime.BprState(sess, P0, Q0, lr=lr, reg=reg, adam_mode=mode)
    loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(j), B).cpu().numpy()
    st.flush()
    om, ol = _oracle_steps(P0, Q0, u, i, j, B, lr, reg, {"lazy": 1, "deferred": 0}[mode])
    np.testing.assert_allclose(loss, ol, rtol=1e-5)
    for name in ("P", "Q", "mP", "mQ"):
        np.testing.assert_allclose(getattr(st, name).cpu().numpy(), getattr(om, name), rtol=0, atol=2e-6, err_msg=name)
    # v = (1 - beta2) g^2 squares the gradient sum; in a batch of tens of thousands the Zipf head's rows add thousands of terms
    # (in a different order on the two sides) and a sum that partly cancels keeps fewer digits: element-wise 1e-4 where v is of
    # the table's size, norm-wise 1e-5 below that
    for name in ("vP", "vQ"):
        ov = getattr(om, name)
        np.testing.assert_allclose(getattr(st, name).cpu().numpy(), ov, rtol=1e-4, atol=1e-5 * float(np.abs(ov).max()), err_msg=name)


def test_runs_beyond_sixteen_bits(sess):
    """A batch of 150 000 triples over 7 users and 5 items: every sorted position, segment start and count beyond 65 535 (the
    descriptor fields' old width), rows cut into thousands of chunks, a front row whose positive slots span thousands of workgroups.
    A row's gradient is a sum of 20 000-60 000 terms of both signs, and long stretches of it are ONE running fp32 sum on every side
    (TF1's segment sum and the C oracle: all of it; here: a front row's tens of thousands of negative slots, a chunked row's
    thousands of chunk partials), good to N * 6e-8 of sum|terms| at worst: no two fp32 implementations agree on such a row beyond
    1e-4.  So the first step is held against the float64 value of the same sums -- |delta g| <= 2e-4 * sum|terms| (seen: 4e-5) --,
    the loss against float64, the step itself to 2e-6 (Adam's m / sqrt(v) does not see the error), and two runs (a ragged second
    batch included) against each other bit for bit."""
    U, I, d, B, lr, reg = 7, 5, 8, 150000, 1e-3, 0.01
    n = B + 1234
    u, i, j = _stream(12, n, U, I)
    g = np.random.Generator(np.random.PCG64(13))
    P0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32)
    Q0 = (g.standard_normal((I, d)) * 0.01).astype(np.float32)
    runs = []
    for _ in range(2):
        st = runtime.BprState(sess, P0, Q0, lr=lr, reg=reg, adam_mode="lazy")
        loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(j), B).cpu().numpy()
        runs.append([loss] + [getattr(st, k).cpu().numpy() for k in ("P", "Q", "mP", "mQ", "vP", "vQ")])
    for x, y in zip(*runs):
        assert np.array_equal(x, y)
    # the first step in float64
    st = runtime.BprState(sess, P0, Q0, lr=lr, reg=reg, adam_mode="lazy")
    loss = st.train_epoch(sess.to_dev(u[:B]), sess.to_dev(i[:B]), sess.to_dev(j[:B]), B).cpu().numpy()
    P, Q = P0.astype(np.float64), Q0.astype(np.float64)
    p, qi, qj = P[u[:B]], Q[i[:B]], Q[j[:B]]
    x = (p * qi).sum(1) - (p * qj).sum(1)
    sg = 1.0 / (1.0 + np.exp(x))  # sigmoid(-x)
    l64 = np.logaddexp(0.0, -x).sum() + reg * 0.5 * ((p * p).sum() + (qi * qi).sum() + (qj * qj).sum())
    assert loss[0] == pytest.approx(l64, rel=1e-6)
    gP, aP, gQ, aQ = np.zeros_like(P), np.zeros_like(P), np.zeros_like(Q), np.zeros_like(Q)
    cu = -sg[:, None] * (qi - qj) + reg * p
    np.add.at(gP, u[:B], cu), np.add.at(aP, u[:B], np.abs(cu))
    for rows, c in ((i[:B], -sg[:, None] * p + reg * qi), (j[:B], sg[:, None] * p + reg * qj)):
        np.add.at(gQ, rows, c), np.add.at(aQ, rows, np.abs(c))
    for name, g64, a64, var0 in (("P", gP, aP, P), ("Q", gQ, aQ, Q)):
        m = getattr(st, "m" + name).cpu().numpy().astype(np.float64)
        worst = float((np.abs(m - 0.1 * g64) / (0.1 * a64 + 1e-30)).max())
        assert worst <= 2e-4, "m%s: |delta| up to %.3g of sum|terms|" % (name, worst)
        v = getattr(st, "v" + name).cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(v, 1e-3 * g64 * g64, rtol=2 * 2e-4 * float((a64 / np.abs(g64)).max()) + 1e-5, err_msg="v" + name)
        lr_t = lr * np.sqrt(1 - 0.999) / (1 - 0.9)
        want = var0 - lr_t * (0.1 * g64) / (np.sqrt(1e-3 * g64 * g64) + 1e-8)
        np.testing.assert_allclose(getattr(st, name).cpu().numpy(), want, rtol=0, atol=2e-6, err_msg=name)


def test_tf1_beyond_8192_is_tf1(sess):
    """model/Recommender.py:17 takes any batch_size.  adam_mode tf1 with batches beyond the in-LDS index runs each call as deferred
    Adam and ends it with the flush (crec_bpr_train_epoch): the tables after a call are TF1's -- against the oracle's TF1 sweep at the
    suite's bounds, and bit for bit what the explicit deferred calls + flush give; two consecutive calls (the second starts from
    rows that are all current, whatever the first left)."""
    U, I, d, B = 900, 700, 16, 9000
    u, i, j = _stream(1, 5 * B + 123, U, I, zipf=1.3)
    g = np.random.Generator(np.random.PCG64(3))
    P0, Q0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32), (g.standard_normal((I, d)) * 0.01).astype(np.float32)
    cut = 2 * B
    du, di, dj = sess.to_dev(u), sess.to_dev(i), sess.to_dev(j)
    out = {}
    for mode in ("tf1", "deferred"):
        st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode=mode)
        loss = [st.train_epoch(du[:cut], di[:cut], dj[:cut], B).cpu().numpy()]
        st.flush()
        loss.append(st.train_epoch(du[cut:], di[cut:], dj[cut:], B).cpu().numpy())
        st.flush()
        out[mode] = [np.concatenate(loss)] + [getattr(st, k).cpu().numpy() for k in ("P", "Q", "mP", "mQ", "vP", "vQ")]
        assert st.adam.step == 6
    for x, y, name in zip(out["tf1"], out["deferred"], ("loss", "P", "Q", "mP", "mQ", "vP", "vQ")):
        assert np.array_equal(x, y), name
    om, ol = _oracle_steps(P0, Q0, u, i, j, B, 1e-3, 0.01, 0)
    np.testing.assert_allclose(out["tf1"][0], ol, rtol=1e-5)
    for dev, name in zip(out["tf1"][1:5], ("P", "Q", "mP", "mQ")):
        np.testing.assert_allclose(dev, getattr(om, name), rtol=0, atol=2e-6, err_msg=name)


def test_tf1_beyond_8192_with_a_slow_beta1_is_refused(sess):
    """(deferred Adam replays idle steps exactly only for beta1 <= 0.9: the refusal names the modes that do run)"""
    U, I, d, B = 100, 80, 16, 9000
    u, i, j = _stream(1, B, U, I)
    P0, Q0 = np.zeros((U, d), np.float32), np.zeros((I, d), np.float32)
    st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1", beta1=0.95)
    with pytest.raises(Exception, match="beta1"):
        st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(j), B)


@pytest.mark.parametrize("mode", ["lazy", "sgd", "adagrad", "deferred"])
def test_big_index_option_changes_nothing(mode):
    U, I, d, B = 20000, 5000, 64, 6144
    n = 3 * B + 55
    u, i, j = _stream(9, n, U, I, zipf=1.25)
    g = np.random.Generator(np.random.PCG64(6))
    P0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32)
    Q0 = (g.standard_normal((I, d)) * 0.01).astype(np.float32)
    out = []
    for big in (0, 1):
        s = runtime.Session(0)
        s.ctx.check(s.lib.crec_ctx_set_option(s.ctx.h, b"big_index", C.c_int64(big)))
        st = runtime.BprState(s, P0, Q0, lr=1e-3, reg=0.01, adam_mode=mode)
        loss = st.train_epoch(s.to_dev(u), s.to_dev(i), s.to_dev(j), B).cpu().numpy()
        st.flush()
        out.append([loss] + [getattr(st, k).cpu().numpy() for k in ("P", "Q", "mP", "mQ", "vP", "vQ")])
        s.close()
    for x, y in zip(*out):
        assert np.array_equal(x, y)


def test_chunks_indexed_ahead_on_a_side_stream_change_nothing():
    """An epoch of several chunks under the `idx_overlap` option: chunk c + 1 is indexed on a side stream beside chunk c's steps (two
    index buffers, chunks of 32, 64, ... batches here).  Same tables and losses, bit for bit, as with every chunk indexed in line
    (the default) -- twice in a row, so that the second epoch's first build meets the first epoch's last steps."""
    U, I, d, B = 30000, 8000, 32, 16384
    n = 75 * B + 123
    u, i, j = _stream(4, n, U, I, zipf=1.25)
    g = np.random.Generator(np.random.PCG64(8))
    P0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32)
    Q0 = (g.standard_normal((I, d)) * 0.01).astype(np.float32)
    out = []
    for overlap in (1, 0):
        s = runtime.Session(0)
        s.ctx.check(s.lib.crec_ctx_set_option(s.ctx.h, b"idx_overlap", C.c_int64(overlap)))
        st = runtime.BprState(s, P0, Q0, lr=1e-3, reg=0.01, adam_mode="lazy")
        a = [s.to_dev(x) for x in (u, i, j)]
        l1 = st.train_epoch(a[0], a[1], a[2], B).cpu().numpy()
        l2 = st.train_epoch(a[0], a[1], a[2], B).cpu().numpy()
        out.append([l1, l2] + [getattr(st, k).cpu().numpy() for k in ("P", "Q", "mP", "mQ", "vP", "vQ")])
        s.close()
    assert len(out[0][0]) == 76
    for x, y in zip(*out):
        assert np.array_equal(x, y)


def test_class_surface_takes_a_large_batch(sess, tmp_path):
    """batch_size = 20000 through the BPR class: validate() lets it through, Adam runs in its deferred form, and the epochs match an
    oracle run of the reference's protocol with TF1's swept Adam (same np.random sampler stream)."""
    import logging

    from cleverrec_amd.data import RankingData
    from cleverrec_amd.model.ranking.BPR import BPR
    g = np.random.Generator(np.random.PCG64(21))
    U, I, B = 700, 400, 20000
    ui_train = {u: [int(x) for x in g.choice(I, size=int(g.integers(10, 50)), replace=False)] for u in range(U)}
    ui_test = {u: [int(x) for x in g.choice(I, size=2, replace=False)] for u in range(0, U, 3)}
    data = RankingData(U, I, ui_train, ui_test)
    configs = {"recommender": "BPR", "epoches": "2", "batch_size": str(B), "test.batch_size": "64", "lr": "0.001",
               "test.neg_samples": "0", "is_pairwise": "True", "init_method": "normal", "stddev": "0.01",
               "loss_func": "bpr", "optimizer": "Adam", "saved_dir": str(tmp_path), "test.interval": "1",
               "topk": "[10,20]", "neg_ratio": "4", "embed_size": "32", "reg": "0.01", "data.split_way": "rs"}
    m = BPR(sess, data, configs, logging.getLogger("bigbatch"))
    assert m.adam_mode == "deferred"
    P0 = (g.standard_normal((U, 32)) * 0.01).astype(np.float32)
    Q0 = (g.standard_normal((I, 32)) * 0.01).astype(np.float32)
    m.build_model(P0, Q0)
    om = oracle.BprModel(P0, Q0, lr=1e-3, reg=0.01)
    np.random.seed(4)
    rng = oracle.Rng(4)
    for _ in range(2):
        avg = m.train_model()
        nb, ou, oi, oj = oracle.sample_pairwise(rng, data.ui_train, data.item_nums, 4, B)
        assert nb >= 3
        oavg, _ = oracle.train_model_py(om, nb, ou, oi, oj, B)
        assert avg == pytest.approx(oavg, rel=1e-5)
    m.state.flush()
    np.testing.assert_allclose(m.P.cpu().numpy(), om.P, rtol=0, atol=2e-6)
    np.testing.assert_allclose(m.Q.cpu().numpy(), om.Q, rtol=0, atol=2e-6)


def test_tf1_on_large_tables_runs_as_deferred_and_is_tf1(sess):
    """Tables whose rows and slots make 64 MB and more: crec_bpr_train_epoch runs adam_mode tf1 as deferred Adam + the flush at the
    end of the call (a sweep per step would cost more than the step).  Against the sweep itself (option tf1_sweep = 1) on the same
    stream: losses to 1e-5, tables and m to 2e-6, v norm-wise 1e-5 -- two consecutive calls; and the launches are the touched-rows
    step's, not the sweep's."""
    U, I, d, B = 150_000, 30_000, 64, 2048
    u, i, j = _stream(5, 7 * B + 99, U, I, zipf=1.2)
    g = np.random.Generator(np.random.PCG64(8))
    P0, Q0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32), (g.standard_normal((I, d)) * 0.01).astype(np.float32)
    du, di, dj = sess.to_dev(u), sess.to_dev(i), sess.to_dev(j)
    cut = 3 * B
    out = {}
    for sweep in (1, 0):
        sess.ctx.set_option("tf1_sweep", sweep)
        sess.ctx.profile(True)
        try:
            st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
            loss = np.concatenate([st.train_epoch(du[:cut], di[:cut], dj[:cut], B).cpu().numpy(), st.train_epoch(du[cut:], di[cut:], dj[cut:], B).cpu().numpy()])
            prof = sess.ctx.profile_read()
        finally:
            sess.ctx.profile(False)
            sess.ctx.set_option("tf1_sweep", 0)
        out[sweep] = [loss] + [getattr(st, k).cpu().numpy() for k in ("P", "Q", "mP", "mQ", "vP", "vQ")]
        lazy, fused = prof.get("bpr_triples_lazy", (0, 0))[1], prof.get("bpr_step_fused", (0, 0))[1] + prof.get("bpr_steps_flow", (0, 0))[1]
        assert (lazy == 0 and fused > 0) if sweep else (lazy > 0 and fused == 0), (sweep, prof)
        assert st.adam.step == 8
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=1e-5)
    for a, b, name in zip(out[0][1:5], out[1][1:5], ("P", "Q", "mP", "mQ")):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-6, err_msg=name)
    for a, b, name in zip(out[0][5:], out[1][5:], ("vP", "vQ")):
        assert np.abs(a.astype(np.float64) - b).max() <= 1e-5 * np.abs(b).max(), name
