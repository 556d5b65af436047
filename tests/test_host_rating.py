"""The FM input reader (cleverrec_amd/model/RatingPreprocess.py) against the reference's own
RatingPreprocess run on the same files (tests/golden/rating_syn.npz, made by tools/make_golden.py),
and -- where the reference checkout is present -- against hashes of its reading of the real
ml-1m.test.libfm.  CPU only."""
import hashlib
import logging
import os

import numpy as np
import pytest

from cleverrec_amd.model.RatingPreprocess import RatingPreprocess


def _cfg(root, name):
    return {"recommender": "FM", "data.root_dir": str(root), "data.dataset": name, "train": ".train.libfm",
            "test": ".test.libfm", "is_real_valued": "True"}


def test_reader_matches_reference(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "rating_syn.npz"))
    d = tmp_path / "syn"
    d.mkdir()
    (d / "syn.train.libfm").write_text("\n".join(g["train_lines"].tolist()) + "\n")
    (d / "syn.test.libfm").write_text("\n".join(g["test_lines"].tolist()) + "\n")
    data = RatingPreprocess(_cfg(tmp_path, "syn"), logging.getLogger("t"))
    assert data.feature_nums == int(g["feature_nums"])
    for mine, ref in ((data.X_idx_tr, "X_idx_tr"), (data.X_value_tr, "X_value_tr"), (data.y_tr, "y_tr"),
                      (data.X_idx_t, "X_idx_t"), (data.X_value_t, "X_value_t"), (data.y_t, "y_t")):
        assert np.array_equal(np.asarray(mine, dtype=np.float64), g[ref].astype(np.float64)), ref


def test_reader_on_the_real_ml1m_test_file(golden_dir):
    path = "/root/reference/dataset/ml-1m/ml-1m.test.libfm"
    if not os.path.exists(path):
        pytest.skip("reference checkout not present (GPU box)")
    g = np.load(os.path.join(golden_dir, "rating_syn.npz"))
    o = RatingPreprocess.__new__(RatingPreprocess)
    o.all_features = {}
    xi, xv, yy = o._read_file(path)
    h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
    assert len(yy) == int(g["ml1m_rows"]) and len(o.all_features) == int(g["ml1m_features"])
    assert h(xi.astype(np.int64)) == str(g["ml1m_idx_hash"])
    assert h(xv.astype(np.float64)) == str(g["ml1m_val_hash"]) and h(yy.astype(np.float64)) == str(g["ml1m_y_hash"])
