"""CML training / rs-evaluation (HIP, through the C ABI) against the torch-autograd restatement of
model/ranking/CML.py with TF1 dense Adam (oracle/models_torch.py).  Needs an MI355X."""
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


@pytest.mark.parametrize("d,K,B", [(16, 5, 200), (128, 20, 200), (16, 3, 20000)])  # 20 000: two chunks of user slots, twenty of item slots
def test_cml_steps_match_oracle(sess, d, K, B):
    g = np.random.Generator(np.random.PCG64(d))
    U, I, N = 70, 90, 2 * B + 50
    P = (g.standard_normal((U, d)) * 0.3).astype(np.float32)
    Q = (g.standard_normal((I, d)) * 0.3).astype(np.float32)
    u, i = g.integers(0, U, N).astype(np.int32), g.integers(0, I, N).astype(np.int32)
    neg = g.integers(0, I, (N, K)).astype(np.int32)
    st = runtime.CmlState(sess, P, Q, lr=1e-3, reg=10.0, margin=1.0)
    om = MT.CML(P, Q, lr=1e-3, reg=10.0, margin=1.0, item_nums=I)
    loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(neg), B).cpu().numpy()
    ol, ol64 = [], []
    for b in range(len(loss)):
        sl = slice(b * B, (b + 1) * B)
        ol64.append(om.loss64(u[sl], i[sl], neg[sl]))  # the same graph in float64 on the oracle's current parameters
        ol.append(om.step(u[sl], i[sl], neg[sl]))
    # The hinge is a difference of squared distances of magnitude ~d * 0.2 against a value of ~1: fp32 rounds it to ~1e-5
    # relative whichever way it is summed.  The bar of north_star (1e-5) is held against the EXACT value of the graph; the
    # fp32 restatement, itself that far from exact, is compared at 2e-5.
    np.testing.assert_allclose(loss, ol64, rtol=1e-5)
    np.testing.assert_allclose(loss, ol, rtol=2e-5)
    assert_adam_close(om, "P", st.P, st.mP, st.vP)
    assert_adam_close(om, "Q", st.Q, st.mQ, st.vQ)


def test_cml_class_surface_and_distance_ranking(sess, tmp_path):
    from cleverrec_amd.model.ranking.CML import CML
    g = np.random.Generator(np.random.PCG64(2))
    U, I, d = 120, 100, 32
    ui_train = {uu: [int(x) for x in g.choice(I, size=int(g.integers(3, 15)), replace=False)] for uu in range(U)}
    ui_test = {uu: [int(x) for x in g.choice(I, size=2, replace=False)] for uu in range(0, U, 2)}
    data = RankingData(U, I, ui_train, ui_test)
    configs = {"recommender": "CML", "epoches": "2", "batch_size": "256", "test.batch_size": "64", "lr": "0.001",
               "test.neg_samples": "0", "is_pairwise": "False", "init_method": "xavier", "stddev": "0.01",
               "loss_func": "hinge", "optimizer": "Adam", "saved_dir": str(tmp_path), "test.interval": "1", "topk": "[10,20]",
               "neg_ratio": "5", "embed_size": str(d), "reg": "10.0", "margin": "1.0", "cml_like": "True", "data.split_way": "rs"}
    m = CML(sess, data, configs, logging.getLogger("t"))
    assert m.cml_like
    init = {"P": (g.standard_normal((U, d)) * 0.4).astype(np.float32), "Q": (g.standard_normal((I, d)) * 0.4).astype(np.float32)}
    m.build_model(init)
    om = MT.CML(init["P"], init["Q"], lr=1e-3, reg=10.0, margin=1.0, item_nums=I)
    np.random.seed(3)
    rng = oracle.Rng(3)
    avg = m.train_model()
    nb, ou, oi, oneg = oracle.sample_cml(rng, data.ui_train, I, 5, 256)
    tot = sum(om.step(ou[b * 256:(b + 1) * 256], oi[b * 256:(b + 1) * 256], oneg[b * 256:(b + 1) * 256]) for b in range(nb))
    assert avg == pytest.approx(tot / nb, rel=2e-5)
    HR, MRR, NDCG = m.test_model_rs()
    dev = MT.CML(m.P.cpu().numpy(), m.Q.cpu().numpy(), lr=1e-3, reg=10.0, margin=1.0, item_nums=I)
    H, M, N = oracle.test_model_rs_py(lambda cur: dev.predict(cur), m.test_users, data.ui_train, data.ui_test, [10, 20], 64,
                                      ascending=True)
    np.testing.assert_allclose(np.mean(NDCG[0]), np.mean(N[0]), rtol=1e-6)
    np.testing.assert_allclose(np.mean(HR[1]), np.mean(H[1]), rtol=1e-6)


def test_cml_loo_uses_unclipped_user_rows(sess, tmp_path):
    """CML._predict on the loo / sampled-negatives branch scores with u_i_dist, built from the UNCLIPPED p_u before
    _unit_clipping rebinds u_embed (CML.py:42,81-82); only the rs branch sees clip_1(p_u).  User rows with |p| > 1 (variables
    are never clipped during training) make the two rankings differ: the candidate ranking must follow the unclipped one."""
    from cleverrec_amd.model.ranking.CML import CML
    g = np.random.Generator(np.random.PCG64(12))
    U, I, d, neg_samples = 80, 300, 32, 40
    ui_train = {uu: [int(x) for x in g.choice(I, size=int(g.integers(3, 15)), replace=False)] for uu in range(U)}
    ui_test = {}
    for uu in range(0, U, 2):  # negatives first, then the held-out items (RankingPreprocess._split_data)
        rest = np.setdiff1d(np.arange(I), ui_train[uu])
        ui_test[uu] = [int(x) for x in g.choice(rest, size=neg_samples + 2, replace=False)]
    data = RankingData(U, I, ui_train, ui_test)
    configs = {"recommender": "CML", "epoches": "1", "batch_size": "256", "test.batch_size": "16", "lr": "0.001",
               "test.neg_samples": str(neg_samples), "is_pairwise": "False", "init_method": "xavier", "stddev": "0.01",
               "loss_func": "hinge", "optimizer": "Adam", "saved_dir": str(tmp_path), "test.interval": "1", "topk": "[5,10]",
               "neg_ratio": "5", "embed_size": str(d), "reg": "10.0", "margin": "1.0", "cml_like": "True", "data.split_way": "loo"}
    m = CML(sess, data, configs, logging.getLogger("t"))
    init = {"P": (g.standard_normal((U, d)) * 0.6).astype(np.float32), "Q": (g.standard_normal((I, d)) * 0.6).astype(np.float32)}
    assert (np.linalg.norm(init["P"], axis=1) > 1.0).all()  # every user row would be clipped on the rs branch
    m.build_model(init)
    HR, MRR, NDCG = m.test_model_loo()
    om = MT.CML(init["P"], init["Q"], lr=1e-3, reg=10.0, margin=1.0, item_nums=I)
    H, M, N = oracle.test_model_loo_py(om.pair_scores, m.test_users, data.ui_test, neg_samples, [5, 10], 16, ascending=True)
    for kid in range(2):
        np.testing.assert_allclose(HR[kid], H[kid], rtol=0)
        np.testing.assert_allclose(NDCG[kid], N[kid], rtol=1e-12)
    # the clipped ranking is a different one on this data (the test would not notice the flag otherwise)
    clipped = lambda u, items: om.predict([u])[0][np.asarray(items)]
    Hc, _, Nc = oracle.test_model_loo_py(clipped, m.test_users, data.ui_test, neg_samples, [5, 10], 16, ascending=True)
    assert not np.allclose(N[1], Nc[1])


@pytest.mark.parametrize("optimizer", ["SGD", "Adagrad"])
def test_cml_sgd_and_adagrad_match_oracle(sess, optimizer):
    """configs['optimizer'] = SGD | Adagrad on the CML graph (utils/tools.py:79-87); both tables are dense variables (the
    covariance term)."""
    from oracle import models_torch as MT
    g = np.random.Generator(np.random.PCG64(23))
    U, I, d, K, B = 60, 45, 16, 5, 100
    N = 2 * B + 17
    P, Q = (g.standard_normal((U, d)) * 0.3).astype(np.float32), (g.standard_normal((I, d)) * 0.3).astype(np.float32)
    u, i = g.integers(0, U, N).astype(np.int32), g.integers(0, I, N).astype(np.int32)
    neg = g.integers(0, I, (N, K)).astype(np.int32)
    lr = 0.01
    st = runtime.CmlState(sess, P, Q, lr=lr, reg=10.0, margin=1.0, optimizer=optimizer)
    om = MT.CML(P, Q, lr=lr, reg=10.0, margin=1.0, item_nums=I, optimizer=optimizer)
    loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(neg), B).cpu().numpy()
    ol = [om.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], neg[b * B:(b + 1) * B]) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=5e-5)
    for dev, name in ((st.P, "P"), (st.Q, "Q")):
        ref = om.numpy(name)
        tol = om.bounds(name)[0] + 1e-6 * np.abs(ref) + 2e-7
        assert (np.abs(dev.cpu().numpy() - ref) <= tol).all(), name
    assert np.abs(st.Q.cpu().numpy() - Q).max() > 1e-3
