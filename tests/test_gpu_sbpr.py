"""Social BPR (model/ranking/SBPR.py) on the GPU against the torch-autograd restatement of its graph with TF1 Adam on
IndexedSlices (oracle/models_torch.py); index streams from the reference-pinned sampler (tests/test_host.py)."""
import logging

import numpy as np
import pytest

import oracle
from oracle import models_torch as MT
from cleverrec_amd import runtime
from cleverrec_amd.data import RankingData
from _adam_close import assert_adam_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sess():
    s = runtime.Session(0)
    yield s
    s.close()


@pytest.mark.parametrize("d,B", [(16, 200), (128, 512), (8, 5461), (8, 5462), (8, 8192), (8, 20000)])  # 3B = 16 383 | 16 386 | 24 576 | 60 000 item slots: one | two | two | four index chunks (and two of user slots at 20 000)
def test_sbpr_steps_match_oracle(sess, d, B):
    g = np.random.Generator(np.random.PCG64(d))
    U, I, N = 80, 60, 2 * B + 37  # a ragged last batch; 3 * B item slots > 2048: several index chunks at B = 512... and one at 200
    P, Q = (g.standard_normal((U, d)) * 0.1).astype(np.float32), (g.standard_normal((I, d)) * 0.1).astype(np.float32)
    bias = (g.standard_normal(I + 1) * 0.05).astype(np.float32)
    u, i, k, j = (g.integers(0, n, N).astype(np.int32) for n in (U, I, I, I))
    suk = g.integers(1, 4, N).astype(np.float32)
    st = runtime.SbprState(sess, P, Q, bias, lr=1e-3, reg=0.05)
    om = MT.SBPR(P, Q, bias, lr=1e-3, reg=0.05)
    loss = st.train_epoch(*(sess.to_dev(x) for x in (u, i, k, j, suk)), B).cpu().numpy()
    ol = [om.step(*(x[b * B:(b + 1) * B] for x in (u, i, k, j, suk))) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=1e-5)
    for name, dev in (("P", st.P), ("Q", st.Q), ("bias", st.bias)):
        assert_adam_close(om, name, dev, *st.slots[name])


def _lazy_pair(sess, g, U, I, d, N, cune):
    P, Q = (g.standard_normal((U, d)) * 0.1).astype(np.float32), (g.standard_normal((I, d)) * 0.1).astype(np.float32)
    bias = (g.standard_normal(I + 1) * 0.05).astype(np.float32)
    u, i, k, j = (g.integers(0, n, N).astype(np.int32) for n in (U, I, I, I))
    suk = g.integers(1, 4, N).astype(np.float32)
    if cune:
        st = runtime.CuneState(sess, P, Q, bias, 0.25, lr=1e-3, reg=0.05, optimizer="LazyAdam")
        om = MT.CUNE_BPR(P, Q, bias, 0.25, lr=1e-3, reg=0.05, optimizer="LazyAdam")
        return st, om, (u, i, k, j), (P, Q, bias)
    st = runtime.SbprState(sess, P, Q, bias, lr=1e-3, reg=0.05, optimizer="LazyAdam")
    om = MT.SBPR(P, Q, bias, lr=1e-3, reg=0.05, optimizer="LazyAdam")
    return st, om, (u, i, k, j, suk), (P, Q, bias)


@pytest.mark.parametrize("cune", [False, True])
def test_lazy_adam_matches_oracle(sess, cune):
    """adam_mode=lazy ('LazyAdam') on SBPR / CUNE_BPR: P, Q rows and bias entries a batch names take Adam's formula, every other
    row / entry and its slots stay bit for bit (CUNE_BPR's scalar s is a dense variable: ApplyAdam as ever).  (a) On the step
    tests' draw (80 users, 60 items, batches of 200): against the torch restatement with the same rule, under its rounding envelope.
    (b) On a population four times the batches' reach: what no batch named keeps its bits; what a batch named is within a hundredth
    of an Adam step on all but a handful of elements (a row touched once has ONE contribution as its gradient, and where -s * (..)
    cancels against reg * p the first Adam step is anybody's: tests/_adam_close.py's loose rule, without the envelope)."""
    B = 200
    st, om, cols, _ = _lazy_pair(sess, np.random.Generator(np.random.PCG64(91 + cune)), 80, 60, 16, 3 * B + 37, cune)
    loss = st.train_epoch(*(sess.to_dev(x) for x in cols), B).cpu().numpy()
    ol = [om.step(*(x[b * B:(b + 1) * B] for x in cols)) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=1e-5)
    for name, dev in (("P", st.P), ("Q", st.Q), ("bias", st.bias)) + ((("s", st.s),) if cune else ()):
        assert_adam_close(om, name, dev, *st.slots[name])

    U, I, B = 500, 400, 128
    st, om, cols, (P, Q, bias) = _lazy_pair(sess, np.random.Generator(np.random.PCG64(93 + cune)), U, I, 32, 3 * B + 19, cune)
    loss = st.train_epoch(*(sess.to_dev(x) for x in cols), B).cpu().numpy()
    ol = [om.step(*(x[b * B:(b + 1) * B] for x in cols)) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=1e-5)
    for name, dev in (("P", st.P), ("Q", st.Q), ("bias", st.bias)):
        diff = np.abs(dev.cpu().numpy().reshape(-1) - om.numpy(name).reshape(-1))
        assert (diff > 1e-5).mean() <= 5e-3 and diff.max() <= 1e-4, "%s: %.3g of the elements beyond 1e-5, max %.3g" % (name, (diff > 1e-5).mean(), diff.max())
    u, i, k, j = cols[:4]
    untouched_u = np.setdiff1d(np.arange(U), u)
    untouched_i = np.setdiff1d(np.arange(I), np.concatenate([i, k, j]))
    assert len(untouched_u) > 20 and len(untouched_i) > 5
    assert np.array_equal(st.P.cpu().numpy()[untouched_u], P[untouched_u])
    assert np.array_equal(st.bias.cpu().numpy().reshape(-1)[untouched_i], bias[untouched_i])
    for name, rows in (("P", untouched_u), ("Q", untouched_i), ("bias", untouched_i)):
        m, v = st.slots[name]
        assert not m.cpu().numpy().reshape(len(m), -1)[rows].any() and not v.cpu().numpy().reshape(len(v), -1)[rows].any(), name


def test_sbpr_more_batches_than_one_item_index_launch_covers(sess):
    """The item-side index is built for 64 batches at a time (sbpr_slots + one rows_index_batches): 68 batches, the last ragged."""
    g = np.random.Generator(np.random.PCG64(68))
    U, I, d, B = 30, 25, 16, 8
    N = 67 * B + 3
    P, Q = (g.standard_normal((U, d)) * 0.1).astype(np.float32), (g.standard_normal((I, d)) * 0.1).astype(np.float32)
    bias = (g.standard_normal(I + 1) * 0.05).astype(np.float32)
    u, i, k, j = (g.integers(0, n, N).astype(np.int32) for n in (U, I, I, I))
    suk = g.integers(1, 4, N).astype(np.float32)
    st = runtime.SbprState(sess, P, Q, bias, lr=1e-3, reg=0.05, optimizer="SGD")  # SGD: 68 steps stay comparable element by element
    om = MT.SBPR(P, Q, bias, lr=1e-3, reg=0.05, optimizer="SGD")
    loss = st.train_epoch(*(sess.to_dev(x) for x in (u, i, k, j, suk)), B).cpu().numpy()
    assert len(loss) == 68
    ol = [om.step(*(x[b * B:(b + 1) * B] for x in (u, i, k, j, suk))) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=1e-5)
    for name, dev in (("P", st.P), ("Q", st.Q), ("bias", st.bias)):
        assert_adam_close(om, name, dev, *st.slots[name])


def test_sbpr_class_surface(sess, tmp_path):
    """SBPR(sess, data, configs, logger): get_SPu, train_model_sbpr on the np.random stream, rs evaluation (u_embed @ Q^T)."""
    from cleverrec_amd.model.ranking.SBPR import SBPR
    g = np.random.Generator(np.random.PCG64(3))
    U, I, d = 90, 120, 32
    ui_train = {uu: [int(x) for x in g.choice(I, size=int(g.integers(3, 12)), replace=False)] for uu in range(U)}
    ui_test = {uu: [int(x) for x in g.choice(I, size=2, replace=False)] for uu in range(0, U, 2)}
    data = RankingData(U, I, ui_train, ui_test)
    data.user_friends = {uu: [int(x) for x in g.choice(U, size=int(g.integers(1, 5)), replace=False) if int(x) != uu]
                         for uu in range(U) if uu % 4 != 3}
    configs = {"recommender": "SBPR", "epoches": "1", "batch_size": "256", "test.batch_size": "64", "lr": "0.001",
               "test.neg_samples": "0", "is_pairwise": "True", "init_method": "normal", "stddev": "0.01", "loss_func": "bpr",
               "optimizer": "Adam", "saved_dir": str(tmp_path), "test.interval": "1", "topk": "[10,20]", "neg_ratio": "3",
               "embed_size": str(d), "reg": "0.05", "data.split_way": "rs"}
    m = SBPR(sess, data, configs, logging.getLogger("t"))
    init = {"P": (g.standard_normal((U, d)) * 0.1).astype(np.float32), "Q": (g.standard_normal((I, d)) * 0.1).astype(np.float32)}
    m.build_model(init)
    om = MT.SBPR(init["P"], init["Q"], np.zeros(I + 1, np.float32), lr=1e-3, reg=0.05)
    from cleverrec_amd.utils import sampler as S
    np.random.seed(7)
    avg = m.train_model()
    np.random.seed(7)
    nb, u, i, k, j, suk = S.ranking_sampler_sbpr(data, m.SPu, 3, 256)
    tot = sum(om.step(*(x[b * 256:(b + 1) * 256] for x in (u, i, k, j, suk))) for b in range(nb))
    assert avg == pytest.approx(tot / nb, rel=1e-5)
    HR, MRR, NDCG = m.test_model_rs()
    Pn, Qn = m.P.cpu().numpy(), m.Q.cpu().numpy()
    H, M, N = oracle.test_model_rs_py(lambda cur: oracle.scores(Pn, Qn, cur, 1), m.test_users, data.ui_train, data.ui_test, [10, 20], 64)
    np.testing.assert_allclose(np.mean(NDCG[0]), np.mean(N[0]), rtol=1e-12)
    np.testing.assert_allclose(np.mean(HR[1]), np.mean(H[1]), rtol=1e-12)


@pytest.mark.parametrize("optimizer", ["Adam", "SGD", "Adagrad"])
@pytest.mark.parametrize("d,B", [(16, 200), (128, 512), (8, 20000)])  # 20 000: the scalar's gradient summed 8192 terms at a time
def test_cune_steps_match_oracle(sess, d, B, optimizer):
    """CUNE_BPR.py:44-61: SBPR's scores, the second log-sigmoid's argument divided by the trainable scalar s + 1."""
    g = np.random.Generator(np.random.PCG64(d + 1))
    U, I, N = 80, 60, 3 * B + 37
    P, Q = (g.standard_normal((U, d)) * 0.1).astype(np.float32), (g.standard_normal((I, d)) * 0.1).astype(np.float32)
    bias = (g.standard_normal(I + 1) * 0.05).astype(np.float32)
    u, i, k, j = (g.integers(0, n, N).astype(np.int32) for n in (U, I, I, I))
    st = runtime.CuneState(sess, P, Q, bias, 0.25, lr=1e-3, reg=0.05, optimizer=optimizer)
    om = MT.CUNE_BPR(P, Q, bias, 0.25, lr=1e-3, reg=0.05, optimizer=optimizer)
    loss = st.train_epoch(*(sess.to_dev(x) for x in (u, i, k, j)), B).cpu().numpy()
    ol = [om.step(*(x[b * B:(b + 1) * B] for x in (u, i, k, j))) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=1e-5)
    for name, dev in (("P", st.P), ("Q", st.Q), ("bias", st.bias), ("s", st.s)):
        assert_adam_close(om, name, dev, *st.slots[name])
    assert float(st.s.cpu()[0]) != pytest.approx(0.25, abs=1e-6)  # s is trained


def test_cune_class_surface(sess, tmp_path):
    """CUNE_BPR(sess, data, configs, logger): network -> walks -> vectors -> friends -> SPu on the host, train_model_sbpr(is_suk=False)
    on the np.random stream, rs evaluation (u_embed @ Q^T), checkpoint names."""
    import torch
    from cleverrec_amd.model.ranking.CUNE_BPR import CUNE_BPR
    from cleverrec_amd.utils import sampler as S
    g = np.random.Generator(np.random.PCG64(5))
    U, I, d = 60, 90, 32
    ui_train = {uu: [int(x) for x in g.choice(I, size=int(g.integers(3, 10)), replace=False)] for uu in range(U) if uu % 9 != 8}
    ui_test = {uu: [int(x) for x in g.choice(I, size=2, replace=False)] for uu in range(0, U, 2) if uu in ui_train}
    data = RankingData(U, I, ui_train, ui_test)
    configs = {"recommender": "CUNE_BPR", "epoches": "1", "batch_size": "256", "test.batch_size": "64", "lr": "0.001",
               "test.neg_samples": "0", "is_pairwise": "True", "init_method": "normal", "stddev": "0.01", "loss_func": "bpr",
               "optimizer": "Adam", "saved_dir": str(tmp_path), "test.interval": "1", "topk": "[10,20]", "neg_ratio": "3",
               "embed_size": str(d), "reg": "0.05", "data.split_way": "rs", "walk_count": "3", "walk_length": "6", "walk_dim": "8",
               "window_size": "3", "topk_f": "4"}
    np.random.seed(11)
    m = CUNE_BPR(sess, data, configs, logging.getLogger("t"))
    assert m.SPu and all(len(f) == 4 and uu not in f for uu, f in data.user_friends.items())
    for uu, lst in m.SPu.items():  # items of the latent friends the user does not hold
        assert not set(lst) & set(ui_train[uu]) and set(lst) <= set().union(*(ui_train[f] for f in data.user_friends[uu]))
    init = {"P": (g.standard_normal((U, d)) * 0.1).astype(np.float32), "Q": (g.standard_normal((I, d)) * 0.1).astype(np.float32)}
    m.build_model(init)
    om = MT.CUNE_BPR(init["P"], init["Q"], np.zeros(I + 1, np.float32), 0.0, lr=1e-3, reg=0.05)
    np.random.seed(7)
    avg = m.train_model()
    np.random.seed(7)
    nb, u, i, k, j = S.ranking_sampler_sbpr(data, m.SPu, 3, 256, is_suk=False)
    tot = sum(om.step(*(x[b * 256:(b + 1) * 256] for x in (u, i, k, j))) for b in range(nb))
    assert avg == pytest.approx(tot / nb, rel=1e-5)
    assert float(m.s.cpu()[0]) == pytest.approx(float(om.numpy("s")[0]), abs=2e-6)
    HR, MRR, NDCG = m.test_model_rs()
    Pn, Qn = m.P.cpu().numpy(), m.Q.cpu().numpy()
    H, M, N = oracle.test_model_rs_py(lambda cur: oracle.scores(Pn, Qn, cur, 1), m.test_users, data.ui_train, data.ui_test, [10, 20], 64)
    np.testing.assert_allclose(np.mean(NDCG[0]), np.mean(N[0]), rtol=1e-12)
    assert sorted(torch.load(m.save())) == ["cune_params/P", "cune_params/Q", "cune_params/bias"]


@pytest.mark.parametrize("neg_samples", [0, 30])
def test_sbpr_loo_ranks_with_the_bias(sess, tmp_path, neg_samples):
    """SBPR._predict on the loo / sampled-negatives branch: pre_scores = ui_scores = p_u.q_i + b_i (SBPR.py:58-59), unlike the rs
    branch (u_embed @ Q^T, no bias).  test_model_loo of the class -- the rows widened by the bias column through the candidate
    top-K -- against oracle.test_model_loo_py fed with the same widened rows (so the fp32 order is the device's)."""
    from cleverrec_amd.model.ranking.SBPR import SBPR
    g = np.random.Generator(np.random.PCG64(9))
    U, I, d = 90, 400, 16
    ui_train = {uu: [int(x) for x in g.choice(I, size=int(g.integers(3, 20)), replace=False)] for uu in range(U)}
    ui_test = {}
    for uu in range(0, U, 2):  # negatives first, then the held-out item(s), like RankingPreprocess._split_data
        rest = np.setdiff1d(np.arange(I), ui_train[uu])
        ui_test[uu] = [int(x) for x in g.choice(rest, size=neg_samples + (1 if neg_samples == 0 else 2), replace=False)]
    data = RankingData(U, I, ui_train, ui_test)
    data.user_friends = {uu: [int(x) for x in g.choice(U, size=3, replace=False) if int(x) != uu] for uu in range(U)}
    configs = {"recommender": "SBPR", "epoches": "1", "batch_size": "256", "test.batch_size": "16", "lr": "0.001",
               "test.neg_samples": str(neg_samples), "is_pairwise": "True", "init_method": "normal", "stddev": "0.01", "loss_func": "bpr",
               "optimizer": "Adam", "saved_dir": str(tmp_path), "test.interval": "1", "topk": "[5,10]", "neg_ratio": "2",
               "embed_size": str(d), "reg": "0.05", "data.split_way": "loo"}
    m = SBPR(sess, data, configs, logging.getLogger("t"))
    init = {"P": (g.standard_normal((U, d)) * 0.3).astype(np.float32), "Q": (g.standard_normal((I, d)) * 0.3).astype(np.float32),
            "bias": (g.standard_normal(I + 1) * 0.5).astype(np.float32)}  # a bias large enough to reorder the items
    m.build_model(init)
    HR, MRR, NDCG = m.test_model_loo()
    Pa = np.concatenate([init["P"], np.ones((U, 1), np.float32), np.zeros((U, 3), np.float32)], 1)
    Qa = np.concatenate([init["Q"], init["bias"][:I, None], np.zeros((I, 3), np.float32)], 1)
    H, M, N = oracle.test_model_loo_py(lambda uu, items: oracle.scores(Pa, Qa[items], [uu], 1)[0], m.test_users, data.ui_test,
                                       neg_samples, [5, 10], 16)
    for kid in range(2):
        np.testing.assert_allclose(HR[kid], H[kid], rtol=0)
        np.testing.assert_allclose(MRR[kid], M[kid], rtol=1e-12)
        np.testing.assert_allclose(NDCG[kid], N[kid], rtol=1e-12)
    if neg_samples:  # and the bias matters: without it the ranking of the candidates is another one
        H0, M0, N0 = oracle.test_model_loo_py(lambda uu, items: oracle.scores(init["P"], init["Q"][items], [uu], 1)[0], m.test_users,
                                              data.ui_test, neg_samples, [5, 10], 16)
        assert abs(np.mean(N0[1]) - np.mean(N[1])) > 1e-3
