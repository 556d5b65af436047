"""The oracle (oracle/) against the reference's own outputs (tests/golden/, made by
tools/make_golden.py from utils/sampler.py, utils/metrics.py) and against numpy's
RandomState, the RNG the reference draws from.  CPU only."""
import json
import os

import numpy as np
import pytest

import oracle


@pytest.mark.parametrize("seed", [0, 1, 123, 2 ** 32 - 1])
def test_rng_stream_matches_numpy(seed):
    r = oracle.Rng(seed)
    np.random.seed(seed)
    ns = [1, 2, 3, 10, 16, 17, 1349, 3706, 4096, 4097, 10 ** 6, 2 ** 31, 2 ** 32 - 1, 2 ** 32, 1, 1, 5]
    for rep in range(40):
        for n in ns:
            assert r.randint(n) == int(np.random.randint(n)), (rep, n)
    for n in [0, 1, 2, 7, 1000, 4097]:
        assert np.array_equal(r.permutation(n), np.random.permutation(n)), n
    # state round trip with numpy (how the host keeps np.random in step)
    st = np.random.get_state()
    r2 = oracle.Rng()
    r2.set_state(st[1], st[2])
    assert [r2.randint(1000) for _ in range(700)] == [int(np.random.randint(1000)) for _ in range(700)]
    key, pos = r2.get_state()
    st2 = np.random.get_state()
    assert np.array_equal(key, st2[1]) and pos == st2[2]


@pytest.mark.parametrize("name", ["toy", "syn200", "dense", "pow2"])
def test_samplers_match_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "sampler_%s.npz" % name))
    ui = oracle.dict_from_csr(g["keys"], g["ptr"], g["items"])
    item_nums, seed, nr, bs = int(g["item_nums"]), int(g["seed"]), int(g["neg_ratio"]), int(g["batch"])

    r = oracle.Rng(seed)
    nb, u, i, j = oracle.sample_pairwise(r, ui, item_nums, nr, bs)
    assert nb == int(g["pw_batches"])
    assert np.array_equal(u, g["pw_u"]) and np.array_equal(i, g["pw_i"]) and np.array_equal(j, g["pw_j"])
    assert r.randint(2 ** 30) == int(g["pw_next"])  # same number of draws consumed

    r = oracle.Rng(seed)
    nb, u, i, y = oracle.sample_pointwise(r, ui, item_nums, nr, bs)
    assert nb == int(g["pt_batches"])
    assert np.array_equal(u, g["pt_u"]) and np.array_equal(i, g["pt_i"]) and np.array_equal(y, g["pt_y"])
    assert r.randint(2 ** 30) == int(g["pt_next"])

    r = oracle.Rng(seed)
    nb, u, i, neg = oracle.sample_cml(r, ui, item_nums, nr, bs)
    assert nb == int(g["cml_batches"])
    assert np.array_equal(u, g["cml_u"]) and np.array_equal(i, g["cml_i"]) and np.array_equal(neg, g["cml_neg"])
    assert r.randint(2 ** 30) == int(g["cml_next"])


def test_toy_anchor_from_survey():
    # SURVEY.md 8(c) toy anchor, measured from the reference
    r = oracle.Rng(123)
    nb, u, i, j = oracle.sample_pairwise(r, {0: [1, 2, 3], 1: [4, 5], 2: [0, 9]}, 10, 2, 4)
    assert nb == 4
    assert u.tolist() == [2, 2, 1, 1, 0, 0, 0, 0, 1, 2, 2, 1, 0, 0]
    assert i.tolist() == [0, 0, 5, 4, 2, 1, 1, 3, 5, 9, 9, 4, 3, 2]
    assert j.tolist() == [1, 4, 0, 0, 0, 9, 6, 0, 3, 7, 3, 9, 9, 6]


def test_metrics_match_reference(golden_dir):
    with open(os.path.join(golden_dir, "metrics.json")) as f:
        g = json.load(f)
    cases = g["ranking"]
    # C oracle, batched: one "user" per case
    Kmax = max(c["K"] for c in cases)
    for c in cases:
        h, m, n = oracle.cal_ranking_metrics_py(c["real"], c["rec"], c["K"])
        assert h == c["hr"] and m == pytest.approx(c["mrr"], abs=1e-15) and n == pytest.approx(c["ndcg"], abs=1e-15)
    ptr = np.zeros(len(cases) + 1, dtype=np.int64)
    ptr[1:] = np.cumsum([len(c["real"]) for c in cases])
    items = np.concatenate([np.array(c["real"], dtype=np.int32) for c in cases])
    for K in sorted({c["K"] for c in cases}):
        sel = [n for n, c in enumerate(cases) if c["K"] == K]
        ids = np.full((len(sel), Kmax), -1, dtype=np.int32)
        for r, n in enumerate(sel):
            ids[r, :K] = cases[n]["rec"]
        hr, mrr, ndcg = oracle.rank_metrics(ids, np.array(sel, dtype=np.int32), [K], ptr, items)
        for r, n in enumerate(sel):
            assert hr[0, r] == cases[n]["hr"]
            assert mrr[0, r] == pytest.approx(cases[n]["mrr"], abs=1e-15)
            assert ndcg[0, r] == pytest.approx(cases[n]["ndcg"], abs=1e-14)


def test_known_metric_answers():
    # SURVEY.md 8(a14) known answers, measured from the reference
    f = oracle.cal_ranking_metrics_py
    assert f([5], [1, 5, 3], 3) == (1.0, 0.5, 0.6309297535714575)
    h, m, n = f([5, 8, 9], [9, 2, 5, 7, 1, 0, 3, 4, 6, 10], 10)
    assert (h, m) == (0.6666666666666666, 1.3333333333333333) and n == pytest.approx(0.7039180890341347, abs=1e-15)
    assert f([1, 2], [7, 8], 2) == (0.0, 0.0, 0.0)
