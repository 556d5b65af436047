"""Device sampler (cleverrec_amd.runtime.DeviceSampler): not stream-compatible with the reference, so it is tested on the
properties the reference's rule guarantees (utils/sampler.py:18-31,50-66,82-93) and against the host sampler's shapes."""
import os

import numpy as np
import pytest
import torch

from cleverrec_amd import runtime
from cleverrec_amd.data import RankingData, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sess():
    s = runtime.Session(0)
    yield s
    s.close()


def _toy(seed, U=300, I=120, deg=30):
    g = np.random.Generator(np.random.PCG64(seed))
    ui_train = {u: g.choice(I, size=int(g.integers(1, deg)), replace=False).tolist() for u in range(U) if u % 7 != 3}
    ui_test = {u: [int(g.integers(0, I))] for u in range(U)}
    return RankingData(U, I, ui_train, ui_test)


def _check_negatives(data, u, neg_rows):
    """every negative is outside the user's training items and a group has no repeats"""
    seen = {k: set(v) for k, v in data.ui_train.items()}
    for uu, row in zip(u, neg_rows):
        assert not (set(row.tolist()) & seen[int(uu)])
        assert len(set(row.tolist())) == len(row)


@pytest.mark.parametrize("nr", [1, 4, 20])
def test_formats_and_rule(sess, nr):
    data = _toy(1)
    ds = runtime.DeviceSampler(sess, data, seed=7)
    n_pos = data.n_train
    neg = ds.negatives(nr).cpu().numpy()
    assert neg.shape == (n_pos, nr) and neg.min() >= 0 and neg.max() < data.item_nums
    eu = ds.edge_user.cpu().numpy()
    _check_negatives(data, eu, neg)
    # pairwise: a permutation of (u, i, j) for every (positive, slot)
    u, i, j = (t.cpu().numpy() for t in ds.pairwise(nr))
    assert u.dtype == i.dtype == j.dtype == np.int32 and len(u) == n_pos * nr
    pos = sorted(zip(np.repeat(eu, nr).tolist(), np.repeat(data.train_csr[2], nr).tolist()))
    assert sorted(zip(u.tolist(), i.tolist())) == pos
    _check_negatives(data, u, j[:, None])
    # pointwise: every positive once with label 1, nr negatives per positive with label 0
    u, i, y = (t.cpu().numpy() for t in ds.pointwise(nr))
    assert len(u) == n_pos * (1 + nr) and y.dtype == np.float32 and int(y.sum()) == n_pos
    assert sorted(zip(u[y == 1].tolist(), i[y == 1].tolist())) == sorted(zip(eu.tolist(), data.train_csr[2].tolist()))
    _check_negatives(data, u[y == 0], i[y == 0][:, None])
    # cml: one row of nr negatives per positive
    u, i, ng = (t.cpu().numpy() for t in ds.cml(nr))
    assert ng.shape == (n_pos, nr)
    assert sorted(zip(u.tolist(), i.tolist())) == sorted(zip(eu.tolist(), data.train_csr[2].tolist()))
    _check_negatives(data, u, ng)


def test_deterministic_per_seed_and_fresh_per_epoch(sess):
    data = _toy(2)
    a, b, c = runtime.DeviceSampler(sess, data, seed=5), runtime.DeviceSampler(sess, data, seed=5), runtime.DeviceSampler(sess, data, seed=6)
    x0, y0, z0 = a.negatives(4).cpu().numpy(), b.negatives(4).cpu().numpy(), c.negatives(4).cpu().numpy()
    assert np.array_equal(x0, y0) and not np.array_equal(x0, z0)
    assert not np.array_equal(x0, a.negatives(4).cpu().numpy())  # the next epoch draws afresh
    pa, pb = a.pairwise(4), b.pairwise(4)  # b is one epoch behind a: different negatives AND different shuffle state
    assert not torch.equal(pa[2], pb[2])


def test_uniform_over_unseen_items(sess):
    """chi-square of the negatives of a user with few training items against the uniform law on its unseen items"""
    I = 64
    data = RankingData(2, I, {0: [0, 1, 2, 3] * 1, 1: list(range(8))}, {0: [5], 1: [9]})
    big = RankingData(2, I, {0: [0, 1, 2, 3], 1: list(range(8))}, {0: [5], 1: [9]})
    ds = runtime.DeviceSampler(sess, big, seed=11)
    counts = np.zeros(I)
    for _ in range(400):
        neg = ds.negatives(8).cpu().numpy()
        counts += np.bincount(neg[:4].reshape(-1), minlength=I)  # the four positives of user 0
    assert counts[:4].sum() == 0
    exp = counts.sum() / (I - 4)
    chi2 = ((counts[4:] - exp) ** 2 / exp).sum()
    assert chi2 < 120  # 59 degrees of freedom: p(chi2 > 120) < 1e-5


def test_too_few_unseen_items_is_an_error(sess):
    data = RankingData(1, 6, {0: [0, 1, 2, 3]}, {0: [4]})
    ds = runtime.DeviceSampler(sess, data, seed=0)
    ds.negatives(2)
    with pytest.raises(Exception):
        ds.negatives(3)


def test_class_surface_with_device_sampler(sess, tmp_path):
    """BPR through the reference's class surface with sampler_mode=device: trains and the loss goes down."""
    import logging
    from cleverrec_amd.model.ranking.BPR import BPR
    data = synthetic("S-ml100k")
    cfg = {"recommender": "BPR", "epoches": "3", "batch_size": "6144", "test.batch_size": "1024", "lr": "0.01", "test.neg_samples": "0",
           "is_pairwise": "True", "init_method": "normal", "stddev": "0.01", "loss_func": "bpr", "optimizer": "Adam",
           "saved_dir": str(tmp_path), "test.interval": "1", "topk": "[10,20]", "neg_ratio": "4", "embed_size": "32", "reg": "0.01",
           "adam_mode": "tf1", "sampler_mode": "device", "seed": "3"}
    m = BPR(sess, data, cfg, logging.getLogger("t"))
    m.build_model()
    l0 = m.train_model()
    for _ in range(4):
        l1 = m.train_model()
    assert np.isfinite(l0) and l1 < l0


# ---- the bit-exact host samplers ON the GPU box, into the device path ----------------------------------
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_rng_streams_match_numpy_on_the_gpu_box(sess):
    """a1: the NumPy-legacy MT19937 randint / permutation of the GPU-bound ctx against numpy itself."""
    ctx = sess.ctx
    for seed in (0, 123, 2 ** 32 - 1):
        ctx.seed(seed)
        np.random.seed(seed)
        for n in [1, 2, 3, 10, 1349, 3706, 4096, 4097, 2 ** 31, 2 ** 32, 1, 7] * 10:
            assert ctx.randint(n) == int(np.random.randint(n))
        for n in (0, 1, 2, 63, 64, 65, 1000, 5000):
            assert np.array_equal(ctx.permutation(n), np.random.permutation(n))


@pytest.mark.parametrize("name", ["toy", "syn200", "dense", "pow2"])
def test_reference_streams_reach_the_device_path_unchanged(sess, name):
    """a2-a4: the pairwise / pointwise / CML samplers of the GPU-bound ctx reproduce the reference-generated streams bit for
    bit (tests/golden/sampler_*.npz, tools/make_golden.py), the columns survive the upload, and a BPR epoch on the pairwise
    stream equals the oracle's on the reference's own columns."""
    import oracle
    g = np.load(os.path.join(GOLDEN, "sampler_%s.npz" % name))
    csr = (g["keys"], g["ptr"], g["items"])
    item_nums, seed, nr = int(g["item_nums"]), int(g["seed"]), int(g["neg_ratio"])
    ctx = sess.ctx
    ctx.seed(seed)
    u, i, j = ctx.sample_pairwise(csr, item_nums, nr)
    assert np.array_equal(u, g["pw_u"]) and np.array_equal(i, g["pw_i"]) and np.array_equal(j, g["pw_j"])
    assert ctx.randint(2 ** 30) == int(g["pw_next"])
    ctx.seed(seed)
    pu, pi, py = ctx.sample_pointwise(csr, item_nums, nr)
    assert np.array_equal(pu, g["pt_u"]) and np.array_equal(pi, g["pt_i"]) and np.array_equal(py, g["pt_y"])
    assert ctx.randint(2 ** 30) == int(g["pt_next"])
    ctx.seed(seed)
    cu, ci, cneg = ctx.sample_cml(csr, item_nums, nr)
    assert np.array_equal(cu, g["cml_u"]) and np.array_equal(ci, g["cml_i"]) and np.array_equal(cneg, g["cml_neg"])
    assert ctx.randint(2 ** 30) == int(g["cml_next"])
    du, di, dj = (sess.to_dev(x) for x in (u, i, j))
    assert np.array_equal(du.cpu().numpy(), g["pw_u"]) and np.array_equal(dj.cpu().numpy(), g["pw_j"])
    # the stream through the step: the device on ITS sampler's columns, the oracle on the reference's
    U = int(max(int(g["keys"].max()), int(g["pw_u"].max())) + 1)
    d, B = 16, 64
    rng = np.random.Generator(np.random.PCG64(1))
    P0, Q0 = (rng.standard_normal((U, d)) * 0.01).astype(np.float32), (rng.standard_normal((item_nums, d)) * 0.01).astype(np.float32)
    st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="tf1")
    loss = st.train_epoch(du, di, dj, B).cpu().numpy()
    om = oracle.BprModel(P0, Q0, lr=1e-3, reg=0.01)
    ru, ri, rj = (np.asarray(g[k], dtype=np.int32) for k in ("pw_u", "pw_i", "pw_j"))
    ol = [om.step(ru[b * B:(b + 1) * B], ri[b * B:(b + 1) * B], rj[b * B:(b + 1) * B]) for b in range(len(loss))]
    np.testing.assert_allclose(loss, ol, rtol=1e-5)
    np.testing.assert_allclose(st.P.cpu().numpy(), om.P, rtol=0, atol=2e-6)
    np.testing.assert_allclose(st.Q.cpu().numpy(), om.Q, rtol=0, atol=2e-6)
