"""cleverrec_amd/model/RankingPreprocess.py against the reference's own RankingPreprocess: the synthetic csv
case of tests/golden/preprocess_syn.npz (tools/make_golden.py) and, where the reference checkout with
ml-100k is present, the hashes of tests/golden/ml100k_anchor.json (preprocess -> samplers).  CPU only."""
import hashlib
import json
import logging
import os

import numpy as np
import pytest

from cleverrec_amd import _lib
from cleverrec_amd.model.RankingPreprocess import RankingPreprocess


def _cfg(root, dataset, fname, sep):
    return {"data.root_dir": str(root), "data.dataset": dataset, "data.file_name": fname, "data.sep": sep,
            "data.format": "UI", "data.split_way": "rs", "data.split_ratio": "[0.7,0.2,0.1]", "data.split_by_time": "False",
            "data.user_min": "0", "data.item_min": "5", "test.neg_samples": "0", "recommender": "BPR"}


def test_preprocess_matches_reference_on_synthetic_csv(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "preprocess_syn.npz"))
    d = tmp_path / "syn"
    d.mkdir()
    with open(d / "ratings.csv", "w") as f:
        for a, b in zip(g["csv_u"], g["csv_i"]):
            f.write("%d,%d\n" % (a, b))
    np.random.seed(int(g["seed"]))
    data = RankingPreprocess(_cfg(tmp_path, "syn", "ratings.csv", ","), logging.getLogger("t"))
    assert (data.user_nums, data.item_nums) == (int(g["user_nums"]), int(g["item_nums"]))
    for mine, ref in zip(data.train_csr, ("train_keys", "train_ptr", "train_items")):
        assert np.array_equal(mine, g[ref]), ref
    for mine, ref in zip(data._test_rows, ("test_keys", "test_ptr", "test_items")):
        assert np.array_equal(mine, g[ref]), ref


def test_preprocess_with_trust_file_matches_reference(golden_dir, tmp_path):
    """social_file (RankingPreprocess.py:49-57): trust pairs whose both ends survive the rating filters, re-indexed, grouped
    into user_friends in file order -- against the reference's own output on a synthetic ratings + trusts pair."""
    g = np.load(os.path.join(golden_dir, "preprocess_social.npz"))
    d = tmp_path / "syn"
    d.mkdir()
    with open(d / "ratings.csv", "w") as f:
        for a, b in zip(g["csv_u"], g["csv_i"]):
            f.write("%d,%d\n" % (a, b))
    with open(d / "trusts.csv", "w") as f:
        for a, b in zip(g["trust_u"], g["trust_v"]):
            f.write("%d,%d\n" % (a, b))
    cfg = dict(_cfg(tmp_path, "syn", "ratings.csv", ","), social_file="trusts.csv", recommender="SBPR")
    np.random.seed(int(g["seed"]))
    data = RankingPreprocess(cfg, logging.getLogger("t"))
    for mine, ref in zip(data.train_csr, ("train_keys", "train_ptr", "train_items")):
        assert np.array_equal(mine, g[ref]), ref
    want = {int(k): g["f_items"][g["f_ptr"][n]:g["f_ptr"][n + 1]].tolist() for n, k in enumerate(g["f_keys"])}
    assert data.user_friends == want and list(data.user_friends.keys()) == list(want.keys())


def test_ml100k_anchor_preprocess_and_samplers(golden_dir):
    root = "/root/reference/dataset"
    if not os.path.exists(os.path.join(root, "ml-100k", "u.data")):
        pytest.skip("reference checkout with ml-100k not present (GPU box)")
    a = json.load(open(os.path.join(golden_dir, "ml100k_anchor.json")))
    h = lambda x: hashlib.sha256(np.ascontiguousarray(x).astype("<i8").tobytes()).hexdigest()[:16]
    np.random.seed(0)
    data = RankingPreprocess(_cfg(root, "ml-100k", "u.data", "\t"), logging.getLogger("t"))
    assert (data.user_nums, data.item_nums, data.n_train) == (a["user_nums"], a["item_nums"], a["n_train"])
    assert len(data.test_users) == a["n_test_users"]
    assert [h(x) for x in data.train_csr] == a["train_hash"]
    ctx = _lib.Context(None)
    ctx.seed(1)
    u, i, j = ctx.sample_pairwise(data.train_csr, data.item_nums, 4)
    assert [h(u), h(i), h(j)] == a["pairwise"]["hash"] and len(u) == a["pairwise"]["n"]
    assert ctx.randint(2 ** 30) == a["pairwise"]["next"]
    ctx.seed(1)
    u, i, y = ctx.sample_pointwise(data.train_csr, data.item_nums, 4)
    assert [h(u), h(i)] == a["pointwise"]["hash"] and float(y.sum()) == a["pointwise"]["ysum"]
    ctx.seed(1)
    u, i, neg = ctx.sample_cml(data.train_csr, data.item_nums, 20)
    assert [h(u), h(i), h(neg)] == a["cml"]["hash"]
