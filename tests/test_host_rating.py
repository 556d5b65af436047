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


def test_libfm_writer_matches_reference_script(golden_dir, tmp_path):
    """utils/fm_to_libfm.py run on a synthetic ratings.dat (tests/golden/libfm_syn.json) vs the writer mirror on
    the same text and seed: both output files byte-identical; and the reader takes the result."""
    import json
    from cleverrec_amd.utils.fm_to_libfm import fm_to_libfm
    g = json.load(open(os.path.join(golden_dir, "libfm_syn.json")))
    d = tmp_path / "syn"
    d.mkdir()
    (d / "ratings.dat").write_text(g["ratings"])
    np.random.seed(g["seed"])
    users, items, n_train, n_test = fm_to_libfm(str(tmp_path), "syn", "ratings.dat")
    assert (d / "syn.train.libfm").read_text() == g["train"]
    assert (d / "syn.test.libfm").read_text() == g["test"]
    assert n_train == len(g["train"].splitlines()) and n_test == len(g["test"].splitlines())
    data = RatingPreprocess(_cfg(tmp_path, "syn"), logging.getLogger("t"))
    assert data.X_idx_tr.shape == (n_train, 2) and data.X_idx_t.shape == (n_test, 2)
    assert data.feature_nums <= users + items and float(data.X_value_tr.min()) == 1.0


def test_libfm_writer_ids_only(golden_dir, tmp_path):
    """is_real_valued=False writes the same rows with bare ids (fm_to_libfm.py:74-76)."""
    import json
    from cleverrec_amd.utils.fm_to_libfm import fm_to_libfm
    g = json.load(open(os.path.join(golden_dir, "libfm_syn.json")))
    d = tmp_path / "syn"
    d.mkdir()
    (d / "ratings.dat").write_text(g["ratings"])
    np.random.seed(g["seed"])
    fm_to_libfm(str(tmp_path), "syn", "ratings.dat", is_real_valued=False)
    assert (d / "syn.train.libfm").read_text() == g["train"].replace(":1", "")
