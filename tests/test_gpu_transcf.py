"""Translational Collaborative Filtering (model/ranking/TransCF.py) on the GPU against the torch-autograd restatement of
its graph (sparse-dense neighbourhood products, hinge, both regularisers) with TF1's dense Adam (oracle/models_torch.py);
index streams from the reference-pinned pairwise sampler (tests/test_host.py)."""
import logging

import numpy as np
import pytest

import oracle
from oracle import models_torch as MT
from cleverrec_amd import runtime
from cleverrec_amd.data import RankingData
from cleverrec_amd.utils.tools import get_sp_mat
from _adam_close import assert_adam_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sess():
    s = runtime.Session(0)
    yield s
    s.close()


def _data(g, U, I, lo=1, hi=12, holes=(7,)):
    """every user but `holes` has lo..hi-1 training items; a few items nobody holds (empty neighbourhoods both ways)"""
    ui_train = {uu: [int(x) for x in g.choice(I - 3, size=int(g.integers(lo, hi)), replace=False)] for uu in range(U) if uu not in holes}
    ui_test = {uu: [int(x) for x in g.choice(I, size=2, replace=False)] for uu in range(0, U, 2)}
    return RankingData(U, I, ui_train, ui_test)


@pytest.mark.parametrize("d,B", [(16, 200), (64, 1100), (128, 97), (16, 20000)])
def test_transcf_steps_match_oracle(sess, d, B):
    """two full batches and a ragged one; 2 * B item slots > 2048 at B = 1100 (several index chunks); users and items
    without neighbours; triples that name such users."""
    g = np.random.Generator(np.random.PCG64(d))
    U, I, N = 80, 60, 2 * B + 37
    data = _data(g, U, I)
    P, Q = (g.standard_normal((U, d)) * 0.3).astype(np.float32), (g.standard_normal((I, d)) * 0.3).astype(np.float32)
    u, i, j = (g.integers(0, n, N).astype(np.int32) for n in (U, I, I))
    st = runtime.TranscfState(sess, P, Q, get_sp_mat(data), lr=1e-3, reg1=0.1, reg2=0.01, margin=0.5)
    om = MT.TransCF(P, Q, data.ui_train, lr=1e-3, reg1=0.1, reg2=0.01, margin=0.5)
    # Batch by batch from the SAME state.  Both tables are dense-Adam variables whose gradient reaches every element, many
    # of them with |g| ~ 1e-7 (sums over neighbourhoods that nearly cancel): Adam's first steps move such an element by
    # anything up to lr depending on the last bits of g, and the next step's gradients of its whole column inherit that --
    # a divergence of the trajectories, not of the step.  The envelope (oracle/models_torch.py) bounds one step.
    import torch
    for b in range(-(-N // B)):
        sl = slice(b * B, (b + 1) * B)
        if b:
            for name, dev in (("P", st.P), ("Q", st.Q)):
                dev.copy_(om.params[name])
                st.slots[name][0].copy_(om.opt.m[name])
                st.slots[name][1].copy_(om.opt.v[name])
            st.adam.beta1_power, st.adam.beta2_power = float(om.opt.b1p), float(om.opt.b2p)
            for k in om.opt.tol_m:  # the states are equal again: the envelope restarts
                om.opt.tol_m[k][:], om.opt.tol_v[k][:], om.opt.tol_var[k][:] = 0.0, 0.0, 0.0
            torch.cuda.synchronize()
        loss = st.train_epoch(*(sess.to_dev(x[sl]) for x in (u, i, j)), B).cpu().numpy()
        assert loss.shape == (1,)
        assert loss[0] == pytest.approx(om.step(u[sl], i[sl], j[sl]), rel=2e-5)
        for name, dev in (("P", st.P), ("Q", st.Q)):
            assert_adam_close(om, name, dev, *st.slots[name])


@pytest.mark.parametrize("optimizer", ["SGD", "Adagrad"])
def test_transcf_sgd_and_adagrad_match_oracle(sess, optimizer):
    """get_optimizer(configs['optimizer'], lr) (utils/tools.py:79-87) on TransCF's graph: both tables take DENSE gradients (the
    neighbourhood products reach every row), so GradientDescentOptimizer and AdagradOptimizer (accumulator 0.1, kept in the m slots)
    move every row every step.  Two full batches and a ragged one in one epoch (SGD / Adagrad do not normalise the gradient: no
    divergence of trajectories to guard against, unlike Adam's first steps), lr large enough to move the tables well beyond the
    comparison's resolution."""
    g = np.random.Generator(np.random.PCG64(71))
    U, I, d, B = 80, 60, 32, 200
    N = 2 * B + 37
    data = _data(g, U, I)
    P, Q = (g.standard_normal((U, d)) * 0.3).astype(np.float32), (g.standard_normal((I, d)) * 0.3).astype(np.float32)
    u, i, j = (g.integers(0, n, N).astype(np.int32) for n in (U, I, I))
    lr = 0.01
    st = runtime.TranscfState(sess, P, Q, get_sp_mat(data), lr=lr, reg1=0.1, reg2=0.01, margin=0.5, optimizer=optimizer)
    om = MT.TransCF(P, Q, data.ui_train, lr=lr, reg1=0.1, reg2=0.01, margin=0.5, optimizer=optimizer)
    loss = st.train_epoch(*(sess.to_dev(x) for x in (u, i, j)), B).cpu().numpy()
    ol = [om.step(*(x[b * B:(b + 1) * B] for x in (u, i, j))) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=2e-5)
    for name, dev in (("P", st.P), ("Q", st.Q)):
        ref = om.numpy(name)
        tol = om.bounds(name)[0] + 1e-6 * np.abs(ref) + 1e-7
        assert (np.abs(dev.cpu().numpy() - ref) <= tol).all(), name
        assert np.abs(ref - (P if name == "P" else Q)).max() > 1e-3  # the steps did move the table


def test_transcf_popular_rows(sess):
    """an item most users hold and a user who holds most items: neighbourhoods longer than one lane group's share (the
    workgroup-per-row launches of csrc/transcf.hip), both in the forward means and in the adjoint sums."""
    g = np.random.Generator(np.random.PCG64(5))
    U, I, d, B = 150, 200, 32, 256
    ui_train = {uu: sorted({0} | {int(x) for x in g.choice(I, size=int(g.integers(2, 9)), replace=False)}) for uu in range(1, U)}
    ui_train[0] = [int(x) for x in g.choice(I, size=130, replace=False)]
    ui_train[3] = ui_train[3][1:]  # one user without the popular item
    data = RankingData(U, I, ui_train, {0: [1, 2]})
    sp = get_sp_mat(data)
    assert np.diff(sp[0]).max() > 64 and np.diff(sp[2]).max() > 64
    P, Q = (g.standard_normal((U, d)) * 0.2).astype(np.float32), (g.standard_normal((I, d)) * 0.2).astype(np.float32)
    u, i, j = (g.integers(0, n, B).astype(np.int32) for n in (U, I, I))
    u[:8], i[:8] = 0, 0  # the long rows are named by the batch
    st = runtime.TranscfState(sess, P, Q, sp, lr=1e-3, reg1=0.1, reg2=0.01, margin=0.5)
    om = MT.TransCF(P, Q, data.ui_train, lr=1e-3, reg1=0.1, reg2=0.01, margin=0.5)
    loss = st.train_epoch(*(sess.to_dev(x) for x in (u, i, j)), B).cpu().numpy()
    assert loss[0] == pytest.approx(om.step(u, i, j), rel=2e-5)
    for name, dev in (("P", st.P), ("Q", st.Q)):
        assert_adam_close(om, name, dev, *st.slots[name])
    users = np.array([0, 3, 149], dtype=np.int32)
    np.testing.assert_allclose(st.scores(sess.to_dev(users)).cpu().numpy(), om.scores(users), rtol=2e-5, atol=1e-6)


def test_transcf_epoch_of_batches(sess):
    """several batches in ONE call (the per-epoch user index, the per-batch item index, a ragged last batch): the losses
    against the oracle's trajectory -- sums over the batch, insensitive to the element-level divergence described above."""
    g = np.random.Generator(np.random.PCG64(77))
    U, I, d, B = 80, 60, 32, 300
    N = 2 * B + 37
    data = _data(g, U, I)
    P, Q = (g.standard_normal((U, d)) * 0.1).astype(np.float32), (g.standard_normal((I, d)) * 0.1).astype(np.float32)
    u, i, j = (g.integers(0, n, N).astype(np.int32) for n in (U, I, I))
    st = runtime.TranscfState(sess, P, Q, get_sp_mat(data), lr=1e-3, reg1=0.1, reg2=0.01, margin=0.5)
    om = MT.TransCF(P, Q, data.ui_train, lr=1e-3, reg1=0.1, reg2=0.01, margin=0.5)
    loss = st.train_epoch(*(sess.to_dev(x) for x in (u, i, j)), B).cpu().numpy()
    ol = [om.step(*(x[b * B:(b + 1) * B] for x in (u, i, j))) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=1e-4)
    assert np.abs(st.P.cpu().numpy() - om.numpy("P")).max() < 2.5e-3  # never more than a couple of Adam steps apart


@pytest.mark.parametrize("d,I,clip", [(16, 70, True), (64, 333, True), (64, 333, False)])
def test_transcf_scores_match_oracle(sess, d, I, clip):
    g = np.random.Generator(np.random.PCG64(11 + d))
    U = 40
    data = _data(g, U, I)
    P, Q = (g.standard_normal((U, d)) * 0.3).astype(np.float32), (g.standard_normal((I, d)) * 0.3).astype(np.float32)
    P[3] *= 5  # rows longer than 1: the clip matters
    st = runtime.TranscfState(sess, P, Q, get_sp_mat(data), lr=1e-3, reg1=0.1, reg2=0.01, margin=0.5)
    om = MT.TransCF(P, Q, data.ui_train, lr=1e-3, reg1=0.1, reg2=0.01, margin=0.5)
    users = np.array([3, 0, 39, 17, 17, 8, 7, 1, 2, 5, 6], dtype=np.int32)  # more than one tile of 8 users; user 7 has no items
    sc = st.scores(sess.to_dev(users), clip_users=clip).cpu().numpy()
    np.testing.assert_allclose(sc, om.scores(users, clip_users=clip), rtol=2e-5, atol=1e-6)


def test_transcf_class_surface(sess, tmp_path):
    """TransCF(sess, data, configs, logger): get_sp_mat, pairwise sampler on the np.random stream, hinge step, rs evaluation
    by ascending distance with the seen filter (cml_like)."""
    from cleverrec_amd.model.ranking.TransCF import TransCF
    g = np.random.Generator(np.random.PCG64(3))
    U, I, d = 90, 120, 32
    data = _data(g, U, I, lo=3, holes=())
    configs = {"recommender": "TransCF", "epoches": "1", "batch_size": "256", "test.batch_size": "64", "lr": "0.001",
               "test.neg_samples": "0", "is_pairwise": "True", "init_method": "normal", "stddev": "0.01", "loss_func": "hinge",
               "optimizer": "Adam", "saved_dir": str(tmp_path), "test.interval": "1", "topk": "[10,20]", "neg_ratio": "4",
               "embed_size": str(d), "margin": "0.5", "reg1": "0.1", "reg2": "0.01", "cml_like": "True", "data.split_way": "rs"}
    m = TransCF(sess, data, configs, logging.getLogger("t"))
    P, Q = (g.standard_normal((U, d)) * 0.3).astype(np.float32), (g.standard_normal((I, d)) * 0.3).astype(np.float32)
    m.build_model({"P": P, "Q": Q})
    om = MT.TransCF(P, Q, data.ui_train, lr=1e-3, reg1=0.1, reg2=0.01, margin=0.5)
    from cleverrec_amd.utils import sampler as S
    np.random.seed(7)
    avg = m.train_model()
    np.random.seed(7)
    train_ = S.pairwise_ranking_sampler(data, 4, 256)
    nb, u, i, j = train_[0], train_[1], train_[2], train_[3]
    tot = sum(om.step(*(x[b * 256:(b + 1) * 256] for x in (u, i, j))) for b in range(nb))
    assert avg == pytest.approx(tot / nb, rel=2e-5)
    HR, MRR, NDCG = m.test_model_rs()
    sc_all = m.state.scores(sess.to_dev(np.arange(U, dtype=np.int32)), clip_users=True).cpu().numpy()
    H, Mr, N = oracle.test_model_rs_py(lambda cur: sc_all[np.asarray(cur)], m.test_users, data.ui_train, data.ui_test, [10, 20], 64,
                                       ascending=True)
    np.testing.assert_allclose(np.mean(NDCG[0]), np.mean(N[0]), rtol=1e-12)
    np.testing.assert_allclose(np.mean(HR[1]), np.mean(H[1]), rtol=1e-12)
    assert m.save().endswith("TransCF.pt")
