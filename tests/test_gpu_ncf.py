"""GMF / MLP / NeuMF training steps (HIP, through the C ABI) against the torch-autograd restatement of the
reference graphs with TF1 Adam semantics (oracle/models_torch.py).  Needs an MI355X.
fp32; losses rtol 1e-5; variables and Adam slots within the propagated fp32 rounding envelope of tests/_adam_close.py
(tight wherever a gradient is well conditioned; a flat atol on post-Adam weights is a cliff edge where a row's
gradient sum cancels)."""
import numpy as np
import pytest
import torch

import oracle
from oracle import models_torch as MT
from cleverrec_amd import runtime
from _adam_close import assert_adam_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sess():
    s = runtime.Session(0)
    yield s
    s.close()


def _instances(seed, U, I, n):
    g = np.random.Generator(np.random.PCG64(seed))
    return (g.integers(0, U, n).astype(np.int32), g.integers(0, I, n).astype(np.int32),
            (g.random(n) < 0.2).astype(np.float32))


def _rand(g, *shape, scale=0.1):
    return (g.standard_normal(shape) * scale).astype(np.float32)


def _tower(g, L0, n_layers):
    Ws, bs, w = [], [], L0
    for _ in range(n_layers):
        Ws.append(_rand(g, w, w // 2, scale=0.2))
        bs.append(_rand(g, w // 2, scale=0.05))
        w //= 2
    return Ws, bs, w


def _slots(st, dev):
    """Adam slots (m, v) of a device variable of an NcfState: a table, or a named view into the flat small block."""
    for k, t in st.t.items():
        if t is dev:
            return st.slots[k]
    off = (dev.data_ptr() - st.small.data_ptr()) // 4
    return st.small_m[off:off + dev.numel()], st.small_v[off:off + dev.numel()]


def _compare(st, om, names, scale=1.0, outliers=0.0):
    for dev, ref in names:
        m, v = _slots(st, dev)
        assert_adam_close(om, ref, dev, m, v, scale=scale, outliers=outliers)


@pytest.mark.parametrize("d", [8, 32, 64])
def test_gmf_steps_match_oracle(sess, d):
    U, I, B = 120, 70, 500
    g = np.random.Generator(np.random.PCG64(d))
    P, Q, h = _rand(g, U, d), _rand(g, I, d), _rand(g, d, scale=0.5)
    u, i, y = _instances(1, U, I, 3 * B + 77)
    st = runtime.NcfState(sess, "GMF", {"P_gmf": P, "Q_gmf": Q}, [], [], h, lr=1e-3, reg_gmf=0.01)
    om = MT.GMF(P, Q, h, lr=1e-3, reg=0.01)
    loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(y), B).cpu().numpy()
    ol = [om.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], y[b * B:(b + 1) * B]) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=1e-5)
    _compare(st, om, [(st.t["P_gmf"], "P"), (st.t["Q_gmf"], "Q"), (st.h, "h_gmf")])


@pytest.mark.parametrize("layers", [[16, 8], [128, 64, 32], [64, 32, 16, 8]])
def test_mlp_steps_match_oracle(sess, layers):
    U, I, B = 90, 60, 300
    g = np.random.Generator(np.random.PCG64(len(layers)))
    L0 = layers[0]
    P, Q = _rand(g, U, L0 // 2), _rand(g, I, L0 // 2)
    Ws, bs, last = _tower(g, L0, len(layers))
    h = _rand(g, last, scale=0.5)
    u, i, y = _instances(2, U, I, 2 * B + 31)
    st = runtime.NcfState(sess, "MLP", {"P_mlp": P, "Q_mlp": Q}, Ws, bs, h, lr=1e-3, reg_mlp=0.01)
    om = MT.MLP(P, Q, Ws, bs, h, lr=1e-3, reg=0.01)
    loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(y), B).cpu().numpy()
    ol = [om.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], y[b * B:(b + 1) * B]) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=1e-5)
    names = [(st.t["P_mlp"], "P"), (st.t["Q_mlp"], "Q"), (st.h, "h_mlp")]
    names += [(st.W[k], "W_%d" % k) for k in range(len(layers))] + [(st.b[k], "b_%d" % k) for k in range(len(layers))]
    _compare(st, om, names)


@pytest.mark.parametrize("d,generic", [(64, 0), (32, 0), (64, 1)])
def test_neumf_config3_shape(sess, d, generic):
    """BASELINE config C3: NeuMF, d = 64, layers [128, 64, 32], batch 6144 (catalogue scaled down); d = 32 is the
    shipped conf.  These tower shapes run shape-specialised tile kernels (csrc/ncf.hip); `ncf_generic` keeps the
    generic kernel under test at the same shape."""
    U, I, B, layers = 400, 300, 6144, [128, 64, 32]
    sess.ctx.set_option("ncf_generic", generic)
    g = np.random.Generator(np.random.PCG64(9))
    Pg, Qg, Pm, Qm = _rand(g, U, d), _rand(g, I, d), _rand(g, U, 64), _rand(g, I, 64)
    Ws, bs, last = _tower(g, 128, 3)
    h = _rand(g, d + last, scale=0.3)
    u, i, y = _instances(3, U, I, 2 * B + 500)
    st = runtime.NcfState(sess, "NeuMF", {"P_gmf": Pg, "Q_gmf": Qg, "P_mlp": Pm, "Q_mlp": Qm}, Ws, bs, h, lr=1e-3,
                          reg_gmf=1e-2, reg_mlp=1e-3)
    om = MT.NeuMF(Pg, Qg, Pm, Qm, Ws, bs, h, lr=1e-3, reg1=1e-2, reg2=1e-3)
    try:
        loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(y), B).cpu().numpy()
    finally:
        sess.ctx.set_option("ncf_generic", 0)
    ol = [om.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], y[b * B:(b + 1) * B]) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=1e-5)
    names = [(st.t[k], k) for k in ("P_gmf", "Q_gmf", "P_mlp", "Q_mlp")] + [(st.h, "h_neumf")]
    names += [(st.W[k], "W_%d" % k) for k in range(3)] + [(st.b[k], "b_%d" % k) for k in range(3)]
    _compare(st, om, names)


def test_neumf_config3_at_the_full_catalogue(sess):
    """BASELINE config C3 at its real size -- NeuMF on the ml-1m catalogue (6040 x 3706), d = 64, layers [128, 64, 32], batch
    6144, instances drawn with the synthetic S-ml1m population's own sampler -- three full batches and a ragged one against
    the torch restatement (NeuMF.py:58-105): losses to 1e-5, the four tables, the tower and the head under the propagated
    rounding envelope.  (test_neumf_config3_shape runs the same tower on a 400 x 300 catalogue, where rows are heavy.)"""
    from cleverrec_amd.data import synthetic
    data = synthetic("S-ml1m")
    U, I, B, d = data.user_nums, data.item_nums, 6144, 64
    g = np.random.Generator(np.random.PCG64(31))
    Pg, Qg, Pm, Qm = _rand(g, U, d), _rand(g, I, d), _rand(g, U, 64), _rand(g, I, 64)
    Ws, bs, last = _tower(g, 128, 3)
    h = _rand(g, d + last, scale=0.3)
    sess.ctx.seed(3)
    u, i, y = sess.ctx.sample_pointwise(data.train_csr, I, 4)
    n = 3 * B + 1234
    u, i, y = u[:n], i[:n], y[:n]
    st = runtime.NcfState(sess, "NeuMF", {"P_gmf": Pg, "Q_gmf": Qg, "P_mlp": Pm, "Q_mlp": Qm}, Ws, bs, h, lr=1e-3,
                          reg_gmf=1e-2, reg_mlp=1e-3)
    om = MT.NeuMF(Pg, Qg, Pm, Qm, Ws, bs, h, lr=1e-3, reg1=1e-2, reg2=1e-3)
    loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(y), B).cpu().numpy()
    assert len(loss) == 4
    ol = [om.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], y[b * B:(b + 1) * B]) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=1e-5)
    names = [(st.t[k], k) for k in ("P_gmf", "Q_gmf", "P_mlp", "Q_mlp")] + [(st.h, "h_neumf")]
    names += [(st.W[k], "W_%d" % k) for k in range(3)] + [(st.b[k], "b_%d" % k) for k in range(3)]
    # At this catalogue a table row meets ONE instance per batch, so its gradient is a single contribution out of the tower's
    # backward pass: nothing averages its rounding, and the envelope's NOISE -- calibrated on rows that sum several contributions --
    # is met by the tail of 1.2 M such elements (seen: 2.03 of the bound on 2 of 386 560 elements of P_mlp, 1.5e-6 absolute; the
    # lazy-Adam tests take 2 for the same reason, tests/_adam_close.py).  Three envelopes (m: 87 of 386 560 elements outside two), and
    # the C5 test's rule for the weights' tail: at most 1e-5 of a table outside, none by more than 2 % of an Adam step; the
    # 1 %-of-a-step census still applies.  The slots' tail moves with the host CPU the restatement runs on (60 elements of m outside
    # three envelopes on another box, each 1-2 ulp of the row's larger elements): _adam_close's `outliers` rule for m and v
    _compare(st, om, names, scale=3.0, outliers=1e-5)


@pytest.mark.parametrize("d,layers", [(32, [16, 8]), (8, [40, 20, 10]), (12, [24])])
def test_neumf_mixed_widths(sess, d, layers):
    """GMF and MLP tables of different row-width classes (two merged-update launches), tower widths that are not
    multiples of the 16 x 16 x 4 MFMA tile (edge blocks, K tails), a ragged last batch and a ragged last tile."""
    U, I, B = 70, 50, 200
    g = np.random.Generator(np.random.PCG64(100 + d))
    L0 = layers[0]
    Pg, Qg, Pm, Qm = _rand(g, U, d), _rand(g, I, d), _rand(g, U, L0 // 2), _rand(g, I, L0 // 2)
    Ws, bs, last = _tower(g, L0, len(layers))
    h = _rand(g, d + last, scale=0.3)
    u, i, y = _instances(4, U, I, 2 * B + 77)
    st = runtime.NcfState(sess, "NeuMF", {"P_gmf": Pg, "Q_gmf": Qg, "P_mlp": Pm, "Q_mlp": Qm}, Ws, bs, h, lr=1e-3,
                          reg_gmf=1e-2, reg_mlp=1e-3)
    om = MT.NeuMF(Pg, Qg, Pm, Qm, Ws, bs, h, lr=1e-3, reg1=1e-2, reg2=1e-3)
    loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(y), B).cpu().numpy()
    ol = [om.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], y[b * B:(b + 1) * B]) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=1e-5)
    names = [(st.t[k], k) for k in ("P_gmf", "Q_gmf", "P_mlp", "Q_mlp")] + [(st.h, "h_neumf")]
    names += [(st.W[k], "W_%d" % k) for k in range(len(layers))] + [(st.b[k], "b_%d" % k) for k in range(len(layers))]
    _compare(st, om, names)


def test_gmf_class_surface_train_and_eval(sess, tmp_path):
    """GMF(sess, data, configs, logger): pointwise train_model() on the np.random stream, then the
    full-catalogue ranking (sigmoid((pu*h).q), GMF.py:40,57) against the oracle's predict()."""
    import logging
    from cleverrec_amd.data import RankingData
    from cleverrec_amd.model.ranking.GMF import GMF
    g = np.random.Generator(np.random.PCG64(4))
    U, I, d = 150, 90, 32
    ui_train = {uu: [int(x) for x in g.choice(I, size=int(g.integers(3, 20)), replace=False)] for uu in range(U)}
    ui_test = {uu: [int(x) for x in g.choice(I, size=2, replace=False)] for uu in range(0, U, 2)}
    data = RankingData(U, I, ui_train, ui_test)
    configs = {"recommender": "GMF", "epoches": "2", "batch_size": "512", "test.batch_size": "64", "lr": "0.001",
               "test.neg_samples": "0", "is_pairwise": "False", "init_method": "xavier_uniform", "stddev": "0.01",
               "loss_func": "cross_entropy", "optimizer": "Adam", "saved_dir": str(tmp_path), "test.interval": "1",
               "topk": "[10,20]", "neg_ratio": "4", "embed_size": str(d), "reg_gmf": "0.01", "data.split_way": "rs"}
    m = GMF(sess, data, configs, logging.getLogger("t"))
    init = {"P": _rand(g, U, d), "Q": _rand(g, I, d), "h_gmf": _rand(g, d, scale=0.5)}
    m.build_model(init)
    om = MT.GMF(init["P"], init["Q"], init["h_gmf"], lr=1e-3, reg=0.01)
    np.random.seed(11)
    rng = oracle.Rng(11)
    for _ in range(2):
        avg = m.train_model()
        nb, ou, oi, oy = oracle.sample_pointwise(rng, data.ui_train, I, 4, 512)
        tot = sum(om.step(ou[b * 512:(b + 1) * 512], oi[b * 512:(b + 1) * 512], oy[b * 512:(b + 1) * 512]) for b in range(nb))
        assert avg == pytest.approx(tot / nb, rel=1e-5)
    HR, MRR, NDCG = m.test_model_rs()
    # same ranking as sorting the oracle's sigmoid scores of the DEVICE parameters (ties aside, none here)
    dev = MT.GMF(m.P.cpu().numpy(), m.Q.cpu().numpy(), m.h_gmf.cpu().numpy(), lr=1e-3, reg=0.01)
    H, M, N = oracle.test_model_rs_py(lambda cur: dev.predict(cur), m.test_users, data.ui_train, data.ui_test, [10, 20], 64)
    np.testing.assert_allclose(np.mean(NDCG[0]), np.mean(N[0]), rtol=1e-6)
    np.testing.assert_allclose(np.mean(HR[1]), np.mean(H[1]), rtol=1e-6)


def test_neumf_full_catalogue_eval(sess, tmp_path):
    """NeuMF(sess, data, configs, logger): test_model_rs() scores every test user against every item through
    the tower (NeuMF.py:62-78) and ranks; compared with the oracle's predict() on the device parameters."""
    import logging
    from cleverrec_amd.data import RankingData
    from cleverrec_amd.model.ranking.NeuMF import NeuMF
    g = np.random.Generator(np.random.PCG64(8))
    U, I, d, layers = 130, 200, 16, [32, 16]
    ui_train = {uu: [int(x) for x in g.choice(I, size=int(g.integers(3, 20)), replace=False)] for uu in range(U)}
    ui_test = {uu: [int(x) for x in g.choice(I, size=2, replace=False)] for uu in range(0, U, 2)}
    data = RankingData(U, I, ui_train, ui_test)
    configs = {"recommender": "NeuMF", "epoches": "1", "batch_size": "512", "test.batch_size": "50", "lr": "0.001",
               "test.neg_samples": "0", "is_pairwise": "False", "init_method": "xavier_uniform", "stddev": "0.01",
               "loss_func": "cross_entropy", "optimizer": "Adam", "saved_dir": str(tmp_path), "test.interval": "1",
               "topk": "[10,20]", "neg_ratio": "4", "embed_size": str(d), "layers": str(layers).replace(" ", ""),
               "reg_gmf": "0.01", "reg_mlp": "0.001", "data.split_way": "rs", "gmf_pretrain": "x", "mlp_pretrain": "y"}
    m = NeuMF(sess, data, configs, logging.getLogger("t"))
    Ws, bs, last = _tower(g, layers[0], len(layers))
    init = {"P_gmf": _rand(g, U, d, scale=0.5), "Q_gmf": _rand(g, I, d, scale=0.5), "P_mlp": _rand(g, U, layers[0] // 2, scale=0.5),
            "Q_mlp": _rand(g, I, layers[0] // 2, scale=0.5), "h_neumf": _rand(g, d + last, scale=0.5)}
    for k in range(len(layers)):
        init["W_%d" % k], init["b_%d" % k] = Ws[k], bs[k]
    m.build_model(init)
    np.random.seed(2)
    m.train_model()
    HR, MRR, NDCG = m.test_model_rs()
    dev = MT.NeuMF(*(m.state.t[k].cpu().numpy() for k in ("P_gmf", "Q_gmf", "P_mlp", "Q_mlp")),
                   [w.cpu().numpy() for w in m.state.W], [b.cpu().numpy() for b in m.state.b], m.h_neumf.cpu().numpy(),
                   lr=1e-3, reg1=0.01, reg2=0.001)
    # logits against the oracle's sigmoid scores (through logit = log(s/(1-s)) would lose bits: compare sigmoids)
    users = sess.to_dev(np.asarray(m.test_users[:40], dtype=np.int32))
    lg = runtime.ncf_scores(sess, m.state, users).cpu().numpy()
    np.testing.assert_allclose(1.0 / (1.0 + np.exp(-lg)), dev.predict(m.test_users[:40]), rtol=2e-5, atol=1e-6)
    H, M, N = oracle.test_model_rs_py(lambda cur: dev.predict(cur), m.test_users, data.ui_train, data.ui_test, [10, 20], 50)
    np.testing.assert_allclose(np.mean(NDCG[0]), np.mean(N[0]), rtol=1e-6)
    np.testing.assert_allclose(np.mean(HR[1]), np.mean(H[1]), rtol=1e-6)


@pytest.mark.parametrize("kind,d", [("NeuMF", 64), ("NeuMF", 32), ("MLP", 0)])
def test_scores_at_the_shipped_tower_shape(sess, kind, d):
    """crec_ncf_scores at layers [128, 64, 32] (shape-specialised ncf_pair_logits, csrc/ncf_eval.hip): against the oracle's
    predict() and against the generic kernel at the same shape (`ncf_generic`); a catalogue that is not a multiple of the
    64-item tile, more users than one workgroup's four."""
    U, I = 37, 333
    g = np.random.Generator(np.random.PCG64(40 + d))
    Ws, bs, last = _tower(g, 128, 3)
    Pm, Qm = _rand(g, U, 64, scale=0.5), _rand(g, I, 64, scale=0.5)
    if kind == "NeuMF":
        Pg, Qg, h = _rand(g, U, d, scale=0.5), _rand(g, I, d, scale=0.5), _rand(g, d + last, scale=0.5)
        st = runtime.NcfState(sess, "NeuMF", {"P_gmf": Pg, "Q_gmf": Qg, "P_mlp": Pm, "Q_mlp": Qm}, Ws, bs, h, lr=1e-3)
        om = MT.NeuMF(Pg, Qg, Pm, Qm, Ws, bs, h, lr=1e-3, reg1=0.0, reg2=0.0)
    else:
        h = _rand(g, last, scale=0.5)
        st = runtime.NcfState(sess, "MLP", {"P_mlp": Pm, "Q_mlp": Qm}, Ws, bs, h, lr=1e-3)
        om = MT.MLP(Pm, Qm, Ws, bs, h, lr=1e-3, reg=0.0)
    users = np.array([0, 36, 5, 5, 17, 2, 9], dtype=np.int32)
    z = runtime.ncf_scores(sess, st, sess.to_dev(users)).cpu().numpy()
    sess.ctx.set_option("ncf_generic", 1)
    try:
        zg = runtime.ncf_scores(sess, st, sess.to_dev(users)).cpu().numpy()
    finally:
        sess.ctx.set_option("ncf_generic", 0)
    np.testing.assert_allclose(z, zg, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(1.0 / (1.0 + np.exp(-z.astype(np.float64))), om.predict(users), rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("kind", ["GMF", "NeuMF"])
def test_square_loss_matches_oracle(sess, kind):
    """loss_func = square: get_loss(.., self.y, self.logits) = sum (y - logits)^2 (utils/tools.py:74-75), the other loss a
    pointwise graph can train with in the reference."""
    U, I, B, d, layers = 90, 70, 256, 16, [32, 16]
    g = np.random.Generator(np.random.PCG64(77))
    u, i, y = _instances(5, U, I, 2 * B + 40)
    if kind == "GMF":
        P, Q, h = _rand(g, U, d), _rand(g, I, d), _rand(g, d, scale=0.5)
        st = runtime.NcfState(sess, "GMF", {"P_gmf": P, "Q_gmf": Q}, [], [], h, lr=1e-3, reg_gmf=0.01, loss_func="square")
        om = MT.GMF(P, Q, h, lr=1e-3, reg=0.01, loss_func="square")
        names = [(st.t["P_gmf"], "P"), (st.t["Q_gmf"], "Q"), (st.h, "h_gmf")]
    else:
        Pg, Qg, Pm, Qm = _rand(g, U, d), _rand(g, I, d), _rand(g, U, layers[0] // 2), _rand(g, I, layers[0] // 2)
        Ws, bs, last = _tower(g, layers[0], len(layers))
        h = _rand(g, d + last, scale=0.3)
        st = runtime.NcfState(sess, "NeuMF", {"P_gmf": Pg, "Q_gmf": Qg, "P_mlp": Pm, "Q_mlp": Qm}, Ws, bs, h, lr=1e-3, reg_gmf=1e-2,
                              reg_mlp=1e-3, loss_func="square")
        om = MT.NeuMF(Pg, Qg, Pm, Qm, Ws, bs, h, lr=1e-3, reg1=1e-2, reg2=1e-3, loss_func="square")
        names = [(st.t[k], k) for k in ("P_gmf", "Q_gmf", "P_mlp", "Q_mlp")] + [(st.h, "h_neumf")]
        names += [(st.W[k], "W_%d" % k) for k in range(len(layers))] + [(st.b[k], "b_%d" % k) for k in range(len(layers))]
    loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(y), B).cpu().numpy()
    ol = [om.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], y[b * B:(b + 1) * B]) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=1e-5)
    _compare(st, om, names)


@pytest.mark.parametrize("optimizer", ["SGD", "Adagrad"])
@pytest.mark.parametrize("kind", ["GMF", "NeuMF"])
def test_sgd_and_adagrad_match_oracle(sess, kind, optimizer):
    """get_optimizer(configs['optimizer'], lr) (utils/tools.py:79-87) on the NCF graphs: GradientDescentOptimizer and
    AdagradOptimizer (accumulator 0.1, kept in the m slots) on the shared row machinery (csrc/rows_adam.hip) -- a sparse
    table's untouched rows stay as they are; the tower's dense variables take the dense forms.  lr large enough that three
    steps move the parameters well beyond the comparison's resolution."""
    U, I, B, d, layers = 90, 70, 256, 16, [32, 16]
    g = np.random.Generator(np.random.PCG64(31))
    u, i, y = _instances(6, U, I, 2 * B + 40)
    lr = 0.05
    if kind == "GMF":
        P, Q, h = _rand(g, U, d), _rand(g, I, d), _rand(g, d, scale=0.5)
        st = runtime.NcfState(sess, "GMF", {"P_gmf": P, "Q_gmf": Q}, [], [], h, lr=lr, reg_gmf=0.01, optimizer=optimizer)
        om = MT.GMF(P, Q, h, lr=lr, reg=0.01, optimizer=optimizer)
        names = [(st.t["P_gmf"], "P"), (st.t["Q_gmf"], "Q"), (st.h, "h_gmf")]
    else:
        Pg, Qg, Pm, Qm = _rand(g, U, d), _rand(g, I, d), _rand(g, U, layers[0] // 2), _rand(g, I, layers[0] // 2)
        Ws, bs, last = _tower(g, layers[0], len(layers))
        h = _rand(g, d + last, scale=0.3)
        st = runtime.NcfState(sess, "NeuMF", {"P_gmf": Pg, "Q_gmf": Qg, "P_mlp": Pm, "Q_mlp": Qm}, Ws, bs, h, lr=lr,
                              reg_gmf=1e-2, reg_mlp=1e-3, optimizer=optimizer)
        om = MT.NeuMF(Pg, Qg, Pm, Qm, Ws, bs, h, lr=lr, reg1=1e-2, reg2=1e-3, optimizer=optimizer)
        names = [(st.t[k], k) for k in ("P_gmf", "Q_gmf", "P_mlp", "Q_mlp")] + [(st.h, "h_neumf")]
        names += [(st.W[k], "W_%d" % k) for k in range(len(layers))] + [(st.b[k], "b_%d" % k) for k in range(len(layers))]
    loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(y), B).cpu().numpy()
    ol = [om.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], y[b * B:(b + 1) * B]) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=1e-5)
    for dev, name in names:
        ref = om.numpy(name)
        tol = om.bounds(name)[0] + 1e-6 * np.abs(ref) + 1e-7
        assert (np.abs(dev.cpu().numpy().reshape(ref.shape) - ref) <= tol).all(), name
    assert np.abs(st.h.cpu().numpy() - h).max() > 1e-3  # the steps did move the parameters
    # untouched rows of a sparse table are bit-for-bit what they were
    t0 = {"GMF": ("P_gmf", "P"), "NeuMF": ("P_gmf", "P_gmf")}[kind]
    untouched = np.setdiff1d(np.arange(U), u)
    if len(untouched):
        init = (P if kind == "GMF" else Pg)[untouched]
        assert np.array_equal(st.t[t0[0]].cpu().numpy()[untouched], init)


def _ncf_pair(sess, kind, g, U, I, d, layers, lr, optimizer):
    """(device state, torch restatement, [(device tensor, restatement name)], first table's (key, initial values))"""
    if kind == "GMF":
        P, Q, h = _rand(g, U, d), _rand(g, I, d), _rand(g, d, scale=0.5)
        st = runtime.NcfState(sess, "GMF", {"P_gmf": P, "Q_gmf": Q}, [], [], h, lr=lr, reg_gmf=0.01, optimizer=optimizer)
        om = MT.GMF(P, Q, h, lr=lr, reg=0.01, optimizer=optimizer)
        return st, om, [(st.t["P_gmf"], "P"), (st.t["Q_gmf"], "Q"), (st.h, "h_gmf")], ("P_gmf", "P", P)
    Pg, Qg, Pm, Qm = _rand(g, U, d), _rand(g, I, d), _rand(g, U, layers[0] // 2), _rand(g, I, layers[0] // 2)
    Ws, bs, last = _tower(g, layers[0], len(layers))
    h = _rand(g, d + last, scale=0.3)
    st = runtime.NcfState(sess, "NeuMF", {"P_gmf": Pg, "Q_gmf": Qg, "P_mlp": Pm, "Q_mlp": Qm}, Ws, bs, h, lr=lr,
                          reg_gmf=1e-2, reg_mlp=1e-3, optimizer=optimizer)
    om = MT.NeuMF(Pg, Qg, Pm, Qm, Ws, bs, h, lr=lr, reg1=1e-2, reg2=1e-3, optimizer=optimizer)
    names = [(st.t[k], k) for k in ("P_gmf", "Q_gmf", "P_mlp", "Q_mlp")] + [(st.h, "h_neumf")]
    names += [(st.W[k], "W_%d" % k) for k in range(len(layers))] + [(st.b[k], "b_%d" % k) for k in range(len(layers))]
    return st, om, names, ("P_gmf", "P_gmf", Pg)


@pytest.mark.parametrize("kind", ["GMF", "NeuMF"])
def test_lazy_adam_matches_oracle(sess, kind):
    """adam_mode=lazy on the NCF graphs (the models' 'LazyAdam'): Adam's IndexedSlices formula on the rows a batch's lookups name,
    every other row of an embedding table and its two slots left exactly as they are (csrc/rows_adam.hip, ROWS_SPARSE_LAZY --
    what BPR's lazy mode does: not TF1's results, whose sparse Adam decays and moves every row at every step); the tower's dense
    variables keep ApplyAdam.  (a) On the draw the SGD / Adagrad test uses: against the torch restatement with the same rule, under
    its rounding envelope.  (b) On a population four times the batches' reach: the rows no batch named keep their bits, and the
    rule parts from TF1's where it should.  (The envelope is not applied to (b): with nearly every row touched once a gradient is
    ONE contribution out of the tower's backward pass, whose inner cancellations the envelope does not model -- TF1 Adam and
    LazyAdam overshoot it alike on such draws, tools/check_lazy_ncf.py.)"""
    layers, lr = [32, 16], 1e-3
    U, I, B, d = 90, 70, 256, 16
    u, i, y = _instances(6, U, I, 2 * B + 40)
    st, om, names, _ = _ncf_pair(sess, kind, np.random.Generator(np.random.PCG64(31)), U, I, d, layers, lr, "LazyAdam")
    loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(y), B).cpu().numpy()
    ol = [om.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], y[b * B:(b + 1) * B]) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=1e-5)
    _compare(st, om, names)

    U, I, B = 300, 200, 128
    u, i, y = _instances(9, U, I, 3 * B + 17)
    st, om, names, (key, okey, init) = _ncf_pair(sess, kind, np.random.Generator(np.random.PCG64(41)), U, I, d, layers, lr, "LazyAdam")
    _, tf1, _, _ = _ncf_pair(sess, kind, np.random.Generator(np.random.PCG64(41)), U, I, d, layers, lr, "Adam")
    loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(y), B).cpu().numpy()
    for b in range(len(loss)):
        sl = slice(b * B, (b + 1) * B)
        assert loss[b] == pytest.approx(om.step(u[sl], i[sl], y[sl]), rel=1e-5)
        tf1.step(u[sl], i[sl], y[sl])
    for dev, name in names:  # flat: 1 % of an Adam step
        np.testing.assert_allclose(dev.cpu().numpy().reshape(om.numpy(name).shape), om.numpy(name), rtol=0, atol=1e-2 * lr, err_msg=name)
    untouched = np.setdiff1d(np.arange(U), u)
    assert len(untouched) > 20
    assert np.array_equal(st.t[key].cpu().numpy()[untouched], init[untouched])
    m, v = st.slots[key]
    assert not m.cpu().numpy()[untouched].any() and not v.cpu().numpy()[untouched].any()
    only_first = np.setdiff1d(u[:B], u[B:])  # touched by the first batch and left alone: TF1 goes on moving them, the lazy rule does not
    assert len(only_first) > 5
    assert np.abs(tf1.numpy(okey)[only_first] - om.numpy(okey)[only_first]).max() > 1e-5


@pytest.mark.parametrize("kind", ["GMF", "NeuMF"])
def test_batch_beyond_one_index_chunk(sess, kind):
    """batch_size = 40 000 (the reference takes any): a batch's user and item slots are indexed 16 384 at a time, three chunks per
    batch here, which the update launch walks in ascending slot order; two full batches and a ragged one against the torch
    restatement under its envelope (the paired P_gmf / P_mlp update of NeuMF included)."""
    U, I, B, d, layers, lr = 2000, 1500, 40000, 16, [32, 16], 1e-3
    u, i, y = _instances(12, U, I, 2 * B + 4321)
    st, om, names, _ = _ncf_pair(sess, kind, np.random.Generator(np.random.PCG64(51)), U, I, d, layers, lr, "Adam")
    loss = st.train_epoch(sess.to_dev(u), sess.to_dev(i), sess.to_dev(y), B).cpu().numpy()
    ol = [om.step(u[b * B:(b + 1) * B], i[b * B:(b + 1) * B], y[b * B:(b + 1) * B]) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=1e-5)
    _compare(st, om, names)
