"""The device form of the reference's samplers (crec_dsampler_*, csrc/mt_sampler.hip): index streams,
the next np.random draw and the generator state bit for bit against the streams the reference itself produced
(tests/golden/sampler_*.npz) and, on random populations, against the oracle's restatement of utils/sampler.py."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from cleverrec_amd import _lib, runtime
from cleverrec_amd.data import RankingData, synthetic

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def sess():
    s = runtime.Session(0)
    yield s
    s.close()


class _Csr:
    def __init__(self, keys, ptr, items, item_nums):
        self.train_csr, self.item_nums = (np.asarray(keys, np.int32), np.asarray(ptr, np.int64), np.asarray(items, np.int32)), int(item_nums)


def _state(seed):
    st = np.random.RandomState(seed).get_state()
    return st[1], st[2]


def _next30(key, pos):
    rs = np.random.RandomState()
    rs.set_state(("MT19937", key, pos, 0, 0.0))
    return int(rs.randint(2 ** 30))


def _np(cols):
    return [c.cpu().numpy() for c in cols]


@pytest.fixture(params=[0, 2, 1], ids=["tables", "four-wave-walk", "general-walk"])
def walk(sess, request):
    """the three forms of the negatives phase (csrc/mt_sampler.hip): per-user tables + look-ups (with the walk behind it for what
    it reports instead of handling), the four-wave walk, the general walk"""
    sess.ctx.set_option("dsampler_general", request.param)
    yield request.param
    sess.ctx.set_option("dsampler_general", 0)


@pytest.mark.parametrize("name", ["toy", "syn200", "dense", "pow2"])
def test_device_samplers_reproduce_the_reference_streams(sess, walk, name):
    """a2-a4 on the GPU: pairwise / pointwise / CML columns and the draw that follows, against the reference's own output"""
    g = np.load(os.path.join(GOLDEN, "sampler_%s.npz" % name))
    ms = runtime.MtSampler(sess, _Csr(g["keys"], g["ptr"], g["items"], g["item_nums"]))
    nr = int(g["neg_ratio"])
    key, pos = _state(int(g["seed"]))
    (u, i, j), end = ms.sample("pairwise", nr, key, pos)
    u, i, j = _np((u, i, j))
    assert u.dtype == i.dtype == j.dtype == np.int32
    assert np.array_equal(u, g["pw_u"]) and np.array_equal(i, g["pw_i"]) and np.array_equal(j, g["pw_j"])
    assert _next30(*end) == int(g["pw_next"])
    (u, i, y), end = ms.sample("pointwise", nr, key, pos)
    u, i, y = _np((u, i, y))
    assert y.dtype == np.float32
    assert np.array_equal(u, g["pt_u"]) and np.array_equal(i, g["pt_i"]) and np.array_equal(y, g["pt_y"])
    assert _next30(*end) == int(g["pt_next"])
    (u, i, ng), end = ms.sample("cml", nr, key, pos)
    u, i, ng = _np((u, i, ng))
    assert np.array_equal(u, g["cml_u"]) and np.array_equal(i, g["cml_i"]) and np.array_equal(ng, g["cml_neg"])
    assert _next30(*end) == int(g["cml_next"])
    ms.close()


def test_a_starved_stream_budget_is_retried_to_the_same_result(sess, walk):
    """the stream budget is sized from expected rejection counts; when an epoch needs more, the walk reports it and the sample is
    redone with twice the words -- same columns, same state (here the first budgets are cut to a fifth on purpose)"""
    g = np.load(os.path.join(GOLDEN, "sampler_syn200.npz"))
    ms = runtime.MtSampler(sess, _Csr(g["keys"], g["ptr"], g["items"], g["item_nums"]))
    key, pos = _state(int(g["seed"]))
    sess.ctx.set_option("dsampler_budget", 20)
    try:
        (u, i, j), end = ms.sample("pairwise", int(g["neg_ratio"]), key, pos)
    finally:
        sess.ctx.set_option("dsampler_budget", 100)
    assert np.array_equal(u.cpu().numpy(), g["pw_u"]) and np.array_equal(j.cpu().numpy(), g["pw_j"])
    assert _next30(*end) == int(g["pw_next"])
    ms.close()


def _population(seed, U, I, lo, hi):
    g = np.random.Generator(np.random.PCG64(seed))
    ui = {}
    for u in range(U):
        if u % 11 == 7:
            continue  # users without train rows
        k = int(g.integers(lo, hi + 1))
        if k == 0:
            continue
        ui[u] = [int(x) for x in g.choice(I, size=min(k, I - 1), replace=False)]
    return RankingData(U, I, ui, {u: [0] for u in range(0, U, 3)})


@pytest.mark.parametrize("U,I,lo,hi,nr,seed", [
    (50, 12, 1, 9, 2, 1),          # tiny catalogue: most candidates rejected, repeats inside groups all the time
    (400, 40, 1, 30, 5, 2),        # dense users, group of 5
    (300, 257, 1, 120, 4, 3),      # item_nums just above a power of two: half of the words fail randint's mask
    (2000, 1349, 5, 300, 4, 4),    # ml-100k-like
    (700, 3000, 1, 2500, 3, 5),    # users that have seen most of the catalogue: thousands of candidates per user
    (1500, 40000, 1, 60, 1, 6),    # bitmap words beyond one per thread (item_nums > 32768), neg_ratio 1
    (3000, 500, 1, 5, 20, 7),      # CML's group of 20 on short lists: several users per tile
    (900, 120, 1, 60, 33, 8),      # a wide group (runtime neg_ratio in the four-wave walk); many repeats
    (200, 90000, 1, 9, 40, 9),     # beyond both of its limits: the general walk either way
    (5000, 64, 0, 3, 3, 10),       # empty users (no train rows) in runs, tiny needs
])
def test_device_samplers_match_the_host_restatement(sess, walk, U, I, lo, hi, nr, seed):
    """random populations: the three samplers' columns and the generator state afterwards against the oracle's Python
    restatement of utils/sampler.py on np.random itself"""
    import oracle
    data = _population(seed, U, I, lo, hi)
    ms = runtime.MtSampler(sess, data)
    for kind in ("pairwise", "pointwise", "cml"):
        np.random.seed(100 + seed)
        np.random.randint(7, size=seed)  # a state in the middle of a block
        st = np.random.get_state()
        cols, end = ms.sample(kind, nr, st[1], st[2])
        cols = _np(cols)
        ref = oracle.reference_sampler(kind, data.ui_train, I, nr)  # draws from np.random
        rst = np.random.get_state()
        for c, r in zip(cols, ref):
                assert np.array_equal(c, np.asarray(r).astype(c.dtype).reshape(c.shape)), kind
        assert end[1] == rst[2] and np.array_equal(end[0], rst[1]), kind
    ms.close()


def test_next_epoch_is_a_drop_in_for_np_random(sess):
    """next_epoch() reads and advances np.random exactly like the reference's sampler; the sample drawn ahead on the side
    stream is used when nobody drew in between and dropped when somebody did"""
    import oracle
    data = _population(11, 600, 300, 2, 40)
    ms = runtime.MtSampler(sess, data)
    np.random.seed(5)
    got = [_np(ms.next_epoch("pairwise", 4, ahead=True)) for _ in range(3)]
    after = np.random.randint(2 ** 30)
    got.append(_np(ms.next_epoch("pairwise", 4, ahead=True)))  # the prefetched sample is stale now: dropped, drawn afresh
    np.random.seed(5)
    ref = [oracle.reference_sampler("pairwise", data.ui_train, 300, 4) for _ in range(3)]
    assert after == np.random.randint(2 ** 30)
    ref.append(oracle.reference_sampler("pairwise", data.ui_train, 300, 4))
    for a, b in zip(got, ref):
        for c, r in zip(a, b):
            assert np.array_equal(c, np.asarray(r).astype(c.dtype))
    ms.drop()
    ms.close()


def test_ml1m_shape_epoch_and_stream_accounting(sess):
    """S-ml1m (BASELINE configs[1]): the device epoch equals the host C++ sampler's (itself pinned to the reference) on 2.8 M
    triples, the state afterwards too; the stream words used add up"""
    data = synthetic("S-ml1m")
    ms = runtime.MtSampler(sess, data)
    key, pos = _state(1)
    (u, i, j), end = ms.sample("pairwise", 4, key, pos)
    ctx = _lib.Context(None)
    ctx.set_state(key, pos)
    hu, hi_, hj = ctx.sample_pairwise(data.train_csr, data.item_nums, 4)
    hkey, hpos = ctx.get_state()
    assert np.array_equal(u.cpu().numpy(), hu) and np.array_equal(i.cpu().numpy(), hi_) and np.array_equal(j.cpu().numpy(), hj)
    assert end[1] == hpos and np.array_equal(end[0], hkey)
    s = ms.stats()
    assert s["candidates_used"] <= s["candidates"] and s["words_negatives"] >= s["candidates_used"] and s["words_shuffle"] >= len(hu) - 1
    assert s["walk_fallbacks"] == 0  # at this shape the parallel negatives phase handles the whole epoch itself
    # the three kinds from one state, then again: deterministic
    a = _np(ms.sample("pointwise", 4, key, pos)[0])
    b = _np(ms.sample("pointwise", 4, key, pos)[0])
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    ms.close()


@pytest.mark.parametrize("model", ["BPR", "GMF", "CML"])
def test_class_surface_trains_identically_on_either_form(sess, tmp_path, model):
    """the drop-in classes draw their epochs on the device by default (sampler_mode=mt19937); with the host form of the same
    stream (mt19937_host) the index streams are the same bit for bit, so the epochs' losses and np.random afterwards are too"""
    import importlib
    import logging
    data = synthetic("S-ml100k")
    cfg = {"recommender": model, "epoches": "3", "batch_size": "6144", "test.batch_size": "1024", "lr": "0.01", "test.neg_samples": "0",
           "is_pairwise": "True" if model != "GMF" else "False", "init_method": "normal", "stddev": "0.01",
           "loss_func": "bpr" if model == "BPR" else "cross_entropy", "optimizer": "Adam", "saved_dir": str(tmp_path), "test.interval": "1",
           "topk": "[10,20]", "neg_ratio": "4" if model != "CML" else "20", "embed_size": "32", "reg": "0.01", "adam_mode": "tf1",
           "data.split_way": "rs", "margin": "1.0", "cml_like": "True" if model == "CML" else "False"}
    cls = getattr(importlib.import_module("cleverrec_amd.model.ranking." + model), model)
    out = {}
    for mode in ("mt19937", "mt19937_host"):
        np.random.seed(77)
        torch.manual_seed(0)
        m = cls(sess, data, dict(cfg, sampler_mode=mode), logging.getLogger("t"))
        assert (m._mt_sampler is not None) == (mode == "mt19937")
        m.build_model()
        losses = [m.train_model() for _ in range(3)]
        m._samplers_done()
        out[mode] = (losses, np.random.randint(2 ** 30))
    assert out["mt19937"][1] == out["mt19937_host"][1]
    np.testing.assert_allclose(out["mt19937"][0], out["mt19937_host"][0], rtol=1e-6)
