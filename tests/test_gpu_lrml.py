"""Latent Relational Metric Learning (model/ranking/LRML.py) on the GPU against the torch-autograd restatement of its
graph with TF1 Adam (IndexedSlices for P, Q; dense for K, M; oracle/models_torch.py); index streams from the
reference-pinned pairwise sampler (tests/test_host.py)."""
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


def _init(g, U, I, d, mem, scale=0.1):
    r = lambda *s: (g.standard_normal(s) * scale).astype(np.float32)
    return r(U, d), r(I, d), r(d, mem) * 3, r(mem, d)  # K larger: attention weights away from uniform


def _slots(st, name):
    d, mem = st.K.shape
    if name in ("P", "Q"):
        return (getattr(st, name),) + st.slots[name]
    m, v = st.slots["small"]
    if name == "K":
        return st.K, m[:d * mem].view(d, mem), v[:d * mem].view(d, mem)
    return st.M, m[d * mem:].view(mem, d), v[d * mem:].view(mem, d)


@pytest.mark.parametrize("d,mem,B,generic", [(16, 8, 200, 0), (64, 50, 300, 0), (128, 50, 1100, 0), (128, 50, 300, 1), (20, 64, 97, 0), (16, 8, 20000, 0)])
def test_lrml_steps_match_oracle(sess, d, mem, B, generic):
    """two full batches and a ragged one (ragged last tile as well); widths that are not multiples of the 16 x 16 x 4 MFMA
    tile (mem = 50, d = 20); 2 * B item slots > 2048 at B = 1100: several index chunks.  128 x 50 (the shipped conf) runs
    shape-specialised kernels; `ncf_generic` keeps the generic ones under test at that shape."""
    sess.ctx.set_option("ncf_generic", generic)
    g = np.random.Generator(np.random.PCG64(d + mem))
    U, I, N = 80, 60, 2 * B + 37
    P, Q, K, M = _init(g, U, I, d, mem)
    u, i, j = (g.integers(0, n, N).astype(np.int32) for n in (U, I, I))
    st = runtime.LrmlState(sess, P, Q, K, M, lr=1e-3, reg=0.001, margin=0.2)
    om = MT.LRML(P, Q, K, M, lr=1e-3, reg=0.001, margin=0.2)
    try:
        loss = st.train_epoch(*(sess.to_dev(x) for x in (u, i, j)), B).cpu().numpy()
    finally:
        sess.ctx.set_option("ncf_generic", 0)
    ol = [om.step(*(x[b * B:(b + 1) * B] for x in (u, i, j))) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=2e-5)
    for name in ("P", "Q", "K", "M"):
        assert_adam_close(om, name, *_slots(st, name))


def test_lrml_lazy_adam_matches_oracle(sess):
    """adam_mode=lazy ('LazyAdam'): P, Q rows a batch names take Adam's sparse formula, every other row and its slots stay bit for
    bit; K, M keep the dense ApplyAdam.  Against the torch restatement with the same rule."""
    g = np.random.Generator(np.random.PCG64(77))
    U, I, d, mem, B = 400, 300, 64, 50, 128
    N = 3 * B + 11
    P, Q, K, M = _init(g, U, I, d, mem)
    u, i, j = (g.integers(0, n, N).astype(np.int32) for n in (U, I, I))
    st = runtime.LrmlState(sess, P, Q, K, M, lr=1e-3, reg=0.001, margin=0.2, optimizer="LazyAdam")
    om = MT.LRML(P, Q, K, M, lr=1e-3, reg=0.001, margin=0.2, optimizer="LazyAdam")
    loss = st.train_epoch(*(sess.to_dev(x) for x in (u, i, j)), B).cpu().numpy()
    ol = [om.step(*(x[b * B:(b + 1) * B] for x in (u, i, j))) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=2e-5)
    for name in ("P", "Q", "K", "M"):
        assert_adam_close(om, name, *_slots(st, name), scale=2.0)
    untouched = np.setdiff1d(np.arange(U), u)
    assert len(untouched) > 20
    var, m, v = _slots(st, "P")
    assert np.array_equal(var.cpu().numpy()[untouched], P[untouched])
    assert not m.cpu().numpy()[untouched].any() and not v.cpu().numpy()[untouched].any()


def test_lrml_hinge_is_active_and_inactive(sess):
    """margin chosen so that a good share of the triples sits on either side of the hinge (the gradient is gated)."""
    g = np.random.Generator(np.random.PCG64(5))
    U, I, d, mem, B = 50, 40, 32, 16, 256
    P, Q, K, M = _init(g, U, I, d, mem, scale=0.3)
    u, i, j = (g.integers(0, n, B).astype(np.int32) for n in (U, I, I))
    om = MT.LRML(P, Q, K, M, lr=1e-3, reg=0.0, margin=0.0)
    w = {k: v for k, v in om.params.items()}
    import torch
    pu, qi, qj = w["P"][torch.as_tensor(u, dtype=torch.long)], w["Q"][torch.as_tensor(i, dtype=torch.long)], w["Q"][torch.as_tensor(j, dtype=torch.long)]
    x = ((pu + om._lram(w, pu, qi) - qi) ** 2).sum(1) - ((pu + om._lram(w, pu, qj) - qj) ** 2).sum(1)
    frac = float((x > 0).float().mean())
    assert 0.2 < frac < 0.8
    st = runtime.LrmlState(sess, P, Q, K, M, lr=1e-3, reg=0.0, margin=0.0)
    loss = st.train_epoch(*(sess.to_dev(a) for a in (u, i, j)), B).cpu().numpy()
    assert loss[0] == pytest.approx(om.step(u, i, j), rel=2e-5)
    for name in ("P", "Q", "K", "M"):
        assert_adam_close(om, name, *_slots(st, name))


@pytest.mark.parametrize("d,mem,I", [(16, 8, 70), (128, 50, 333)])
def test_lrml_scores_match_oracle(sess, d, mem, I):
    g = np.random.Generator(np.random.PCG64(11 + d))
    U = 40
    P, Q, K, M = _init(g, U, I, d, mem, scale=0.3)
    st = runtime.LrmlState(sess, P, Q, K, M, lr=1e-3, reg=0.001, margin=0.2)
    om = MT.LRML(P, Q, K, M, lr=1e-3, reg=0.001, margin=0.2)
    users = np.array([3, 0, 39, 17, 17, 8], dtype=np.int32)
    sc = st.scores(sess.to_dev(users)).cpu().numpy()
    np.testing.assert_allclose(sc, om.scores(users), rtol=2e-5)


def test_lrml_class_surface(sess, tmp_path):
    """LRML(sess, data, configs, logger): pairwise sampler on the np.random stream, hinge step, rs evaluation by ascending
    distance with the seen filter (cml_like)."""
    from cleverrec_amd.model.ranking.LRML import LRML
    g = np.random.Generator(np.random.PCG64(3))
    U, I, d, mem = 90, 120, 32, 10
    ui_train = {uu: [int(x) for x in g.choice(I, size=int(g.integers(3, 12)), replace=False)] for uu in range(U)}
    ui_test = {uu: [int(x) for x in g.choice(I, size=2, replace=False)] for uu in range(0, U, 2)}
    data = RankingData(U, I, ui_train, ui_test)
    configs = {"recommender": "LRML", "epoches": "1", "batch_size": "256", "test.batch_size": "64", "lr": "0.001",
               "test.neg_samples": "0", "is_pairwise": "True", "init_method": "normal", "stddev": "0.01", "loss_func": "hinge",
               "optimizer": "Adam", "saved_dir": str(tmp_path), "test.interval": "1", "topk": "[10,20]", "neg_ratio": "1",
               "embed_size": str(d), "mem_size": str(mem), "margin": "0.2", "reg": "0.001", "cml_like": "True",
               "data.split_way": "rs"}
    m = LRML(sess, data, configs, logging.getLogger("t"))
    P, Q, K, M = _init(g, U, I, d, mem, scale=0.3)
    m.build_model({"P": P, "Q": Q, "K": K, "M": M})
    om = MT.LRML(P, Q, K, M, lr=1e-3, reg=0.001, margin=0.2)
    from cleverrec_amd.utils import sampler as S
    np.random.seed(7)
    avg = m.train_model()
    np.random.seed(7)
    train_ = S.pairwise_ranking_sampler(data, 1, 256)
    nb, u, i, j = train_[0], train_[1], train_[2], train_[3]
    tot = sum(om.step(*(x[b * 256:(b + 1) * 256] for x in (u, i, j))) for b in range(nb))
    assert avg == pytest.approx(tot / nb, rel=2e-5)
    HR, MRR, NDCG = m.test_model_rs()
    # the metrics of the device's own distances under the reference's argsort + seen-filter loop: the top-K / filter path;
    # the distances themselves are checked against float64 in test_lrml_scores_match_oracle
    sc_all = m.state.scores(sess.to_dev(np.arange(U, dtype=np.int32))).cpu().numpy()
    H, Mr, N = oracle.test_model_rs_py(lambda cur: sc_all[np.asarray(cur)], m.test_users, data.ui_train, data.ui_test, [10, 20], 64,
                                       ascending=True)
    np.testing.assert_allclose(np.mean(NDCG[0]), np.mean(N[0]), rtol=1e-12)
    np.testing.assert_allclose(np.mean(HR[1]), np.mean(H[1]), rtol=1e-12)
    assert m.save().endswith("LRML.pt")
