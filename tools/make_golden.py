#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by RUNNING the reference's own pure-Python
hot-path code (utils/sampler.py, utils/metrics.py, model/RankingPreprocess.py).

Run in the build container only (needs /root/reference); the outputs are data
(inputs + expected outputs) and are committed, the reference code is not.

utils/tools.py imports tensorflow and gensim at module top although the
sampler / preprocess only use its `timer` and `re_index`; both packages are
absent here, so inert empty modules are registered under those names before the
import.  No reference file is modified or copied.

    python tools/make_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import hashlib
import json
import logging
import os
import sys
import tempfile
import types

import numpy as np


def _import_reference(ref):
    sys.dont_write_bytecode = True
    sys.path.insert(0, ref)
    for name in ("tensorflow", "gensim", "gensim.models", "gensim.models.word2vec"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["gensim"].models = sys.modules["gensim.models"]
    sys.modules["gensim.models"].word2vec = sys.modules["gensim.models.word2vec"]
    import utils.metrics as metrics
    import utils.sampler as sampler
    return sampler, metrics


class Data:
    pass


def synth_ui(seed, n_users, n_items, lo, hi, skip_every=0):
    """Small synthetic ui_train in the reference's protocol: dict[int -> list[int]],
    keys ascending, item lists unsorted without duplicates."""
    g = np.random.Generator(np.random.PCG64(seed))
    pop = 1.0 / np.arange(1, n_items + 1)
    pop /= pop.sum()
    ui = {}
    for u in range(n_users):
        if skip_every and u % skip_every == skip_every - 1:
            continue  # users without train rows exist in the reference protocol
        k = int(g.integers(lo, hi + 1))
        ui[u] = [int(x) for x in g.choice(n_items, size=k, replace=False, p=pop)]
    return ui


def csr(ui):
    keys = np.array(list(ui.keys()), dtype=np.int32)
    ptr = np.zeros(len(ui) + 1, dtype=np.int64)
    ptr[1:] = np.cumsum([len(v) for v in ui.values()])
    items = np.array([x for v in ui.values() for x in v], dtype=np.int32)
    return keys, ptr, items


def sampler_case(sampler, name, ui, item_nums, seed, neg_ratio, batch):
    d = Data()
    d.ui_train, d.item_nums = ui, item_nums
    out = {}
    keys, ptr, items = csr(ui)
    out.update(keys=keys, ptr=ptr, items=items, item_nums=item_nums, seed=seed, neg_ratio=neg_ratio, batch=batch)
    np.random.seed(seed)
    nb, u, i, j = sampler.pairwise_ranking_sampler(d, neg_ratio, batch)
    out.update(pw_batches=nb, pw_u=u.astype(np.int64), pw_i=i.astype(np.int64), pw_j=j.astype(np.int64),
               pw_next=np.random.randint(2 ** 30))
    np.random.seed(seed)
    nb, u, i, y = sampler.pointwise_ranking_sampler(d, neg_ratio, batch)
    out.update(pt_batches=nb, pt_u=u.astype(np.int64), pt_i=i.astype(np.int64), pt_y=y.astype(np.float64),
               pt_next=np.random.randint(2 ** 30))
    np.random.seed(seed)
    nb, u, i, neg = sampler.ranking_sampler_cml(d, neg_ratio, batch)
    out.update(cml_batches=nb, cml_u=u.astype(np.int64), cml_i=i.astype(np.int64), cml_neg=neg.astype(np.int64),
               cml_next=np.random.randint(2 ** 30))
    return name, out


def sbpr_case(sampler, seed=5, n_users=60, n_items=80, neg_ratio=3, batch=64):
    """ranking_sampler_sbpr (utils/sampler.py:102-141) with get_SPu (utils/tools.py:116-127) on a synthetic trust graph:
    inputs (ui_train, user_friends), the reference's SPu lists IN THEIR LIST ORDER (a Python set turned into a list),
    the five shuffled columns and the next np.random draw."""
    import utils.tools as tools
    g = np.random.Generator(np.random.PCG64(seed))
    ui = synth_ui(seed, n_users, n_items, 2, 9, skip_every=7)
    friends = {}
    for u in range(n_users):
        if u % 5 == 4:
            continue  # users without friends
        k = int(g.integers(1, 6))
        friends[u] = [int(x) for x in g.choice(n_users, size=k, replace=False) if int(x) != u]
        if not friends[u]:
            del friends[u]
    d = Data()
    d.ui_train, d.item_nums, d.user_friends = ui, n_items, friends
    SPu = tools.get_SPu(d)
    np.random.seed(seed)
    nb, u, i, i_s, i_neg, suk = sampler.ranking_sampler_sbpr(d, SPu, neg_ratio, batch)
    nxt = np.random.randint(2 ** 30)
    keys, ptr, items = csr(ui)
    fk, fp, fi = csr(friends)
    sk, sp, si = csr(SPu)
    return dict(keys=keys, ptr=ptr, items=items, item_nums=n_items, f_keys=fk, f_ptr=fp, f_items=fi, spu_keys=sk, spu_ptr=sp,
                spu_items=si, seed=seed, neg_ratio=neg_ratio, batch=batch, batches=nb, u=u.astype(np.int64), i=i.astype(np.int64),
                i_s=i_s.astype(np.int64), i_neg=i_neg.astype(np.int64), suk=suk.astype(np.int64), next=nxt)


def sp_mat_case(seed=9, n_users=70, n_items=50):
    """get_sp_mat (utils/tools.py:100-113, TransCF's two neighbourhood matrices): the index / value lists the reference's own
    loops hand to tf.SparseTensor -- the inert tensorflow module gets a SparseTensor attribute that returns its arguments --
    for a synthetic ui_train with users without rows and items nobody holds."""
    import tensorflow as tf
    import utils.tools as tools
    tf.SparseTensor = lambda indices, values, dense_shape: (indices, values, dense_shape)
    try:
        ui = synth_ui(seed, n_users, n_items, 1, 12, skip_every=9)
        d = Data()
        d.ui_train, d.user_nums, d.item_nums = ui, n_users, n_items
        (ui_idx, ui_val, ui_shape), (iu_idx, iu_val, iu_shape) = tools.get_sp_mat(d)
    finally:
        del tf.SparseTensor
    keys, ptr, items = csr(ui)
    return dict(keys=keys, ptr=ptr, items=items, user_nums=n_users, item_nums=n_items, ui_indices=np.array(ui_idx, dtype=np.int64),
                ui_values=np.array(ui_val, dtype=np.float64), ui_shape=np.array(ui_shape), iu_indices=np.array(iu_idx, dtype=np.int64),
                iu_values=np.array(iu_val, dtype=np.float64), iu_shape=np.array(iu_shape))


def cune_vectors(tokens, dim, seed=77):
    """Deterministic stand-in for the word vectors (a function of the token alone, independent of np.random): what the
    recorder below hands the reference, and what tests/test_host.py hands the repo's own top-K code."""
    out = {}
    for t in tokens:
        g = np.random.Generator(np.random.PCG64([seed, int(t)]))
        out[t] = g.standard_normal(dim).astype(np.float32)
    return out


def cune_case(seed=13, n_users=45, n_items=35, walk_count=3, walk_length=8, walk_dim=6, window=3, topk_f=4):
    """get_topK_friends_and_SPu (utils/tools.py:130-207, CUNE_BPR) run by the reference itself: the user network, the deep
    walks on the np.random stream, the cosine top-K friends and SPu are the reference's own loops.  gensim is absent; the
    inert gensim.models.word2vec module gets a Word2Vec attribute that RECORDS the walks it is given and answers `.wv[token]`
    with cune_vectors (so the walks are pinned exactly, and friends / SPu are pinned given those vectors)."""
    import utils.tools as tools
    seen = {}

    class Recorder:
        def __init__(self, walks, size=None, window=None, min_count=None, iter=None):
            seen["walks"], seen["args"] = [list(w) for w in walks], (size, window, min_count, iter)
            self.wv = cune_vectors(sorted({t for w in walks for t in w}, key=int), size)
    tools.word2vec.Word2Vec = Recorder
    try:
        ui = synth_ui(seed, n_users, n_items, 1, 6, skip_every=8)
        d = Data()
        d.ui_train, d.item_nums, d.user_nums = ui, n_items, n_users
        np.random.seed(seed)
        friends, SPu = tools.get_topK_friends_and_SPu(d, walk_count, walk_length, walk_dim, window, topk_f)
        nxt = np.random.randint(2 ** 30)
    finally:
        del tools.word2vec.Word2Vec
    keys, ptr, items = csr(ui)
    fk, fp, fi = csr({u: [int(x) for x in f] for u, f in friends.items()})
    sk, sp, si = csr(SPu)
    return dict(keys=keys, ptr=ptr, items=items, item_nums=n_items, user_nums=n_users, seed=seed,
                params=np.array([walk_count, walk_length, walk_dim, window, topk_f]), w2v_args=np.array(seen["args"]),
                walks=np.array([[int(t) for t in w] for w in seen["walks"]], dtype=np.int64), f_keys=fk, f_ptr=fp, f_items=fi,
                spu_keys=sk, spu_ptr=sp, spu_items=si, next=nxt)


def metrics_cases(metrics, n=400):
    g = np.random.Generator(np.random.PCG64(7))
    cases = [([5], [1, 5, 3], 3), ([5, 8, 9], [9, 2, 5, 7, 1, 0, 3, 4, 6, 10], 10), ([1, 2], [7, 8], 2),
             ([3, 7], [7, 1, 3, 9], 4)]
    for _ in range(n):
        n_items = int(g.integers(5, 60))
        K = int(g.integers(1, min(20, n_items) + 1))
        n_real = int(g.integers(1, min(15, n_items) + 1))
        real = [int(x) for x in g.choice(n_items, size=n_real, replace=False)]
        rec = [int(x) for x in g.choice(n_items, size=K, replace=False)]
        cases.append((real, rec, K))
    out = []
    for real, rec, K in cases:
        # the reference hands rec_items over as a float64 ndarray slice (RankingRecommender.py:229,242)
        hr, mrr, ndcg = metrics.cal_ranking_metrics(real, np.array(rec, dtype=np.float64), K)
        out.append({"real": real, "rec": rec, "K": K, "hr": float(hr), "mrr": float(mrr), "ndcg": float(ndcg)})
    rm = metrics.cal_rmse_mae([1, 2, 3], [1.5, 2, 1])
    return {"ranking": out, "rmse_mae": {"y": [1, 2, 3], "y_pre": [1.5, 2, 1], "rmse": rm[0], "mae": rm[1]}}


def preprocess_case(ref, seed=0, social=False):
    """RankingPreprocess on a synthetic UI csv (first row is swallowed as a header,
    SURVEY 2.3) -> the `data` protocol the hot path consumes.  social: plus a trust file (RankingPreprocess.py:49-57):
    pairs over a wider id range than the ratings', so that the invalid-user filter and the re-indexing are exercised."""
    from model.RankingPreprocess import RankingPreprocess
    g = np.random.Generator(np.random.PCG64(11))
    rows = set()
    while len(rows) < 3000:
        rows.add((int(g.integers(100, 260)), int(g.zipf(1.3)) % 400 + 1000))
    rows = sorted(rows, key=lambda r: (r[0] * 7919 + r[1] * 104729) % 65521)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "syn"))
    with open(os.path.join(tmp, "syn", "ratings.csv"), "w") as f:
        for a, b in rows:
            f.write("%d,%d\n" % (a, b))
    cfg = {"data.root_dir": tmp, "data.dataset": "syn", "data.file_name": "ratings.csv", "data.sep": ",",
           "data.format": "UI", "data.split_way": "rs", "data.split_ratio": "[0.7,0.2,0.1]",
           "data.split_by_time": "False", "data.user_min": "0", "data.item_min": "5", "test.neg_samples": "0",
           "recommender": "BPR"}
    logger = logging.getLogger("golden")
    extra = {}
    if social:
        pairs = [(int(g.integers(90, 280)), int(g.integers(90, 280))) for _ in range(900)]
        with open(os.path.join(tmp, "syn", "trusts.csv"), "w") as f:
            for a, b in pairs:
                f.write("%d,%d\n" % (a, b))
        cfg["social_file"] = "trusts.csv"
        cfg["recommender"] = "SBPR"
        extra.update(trust_u=np.array([p[0] for p in pairs], dtype=np.int64), trust_v=np.array([p[1] for p in pairs], dtype=np.int64))
    np.random.seed(seed)
    d = RankingPreprocess(cfg, logger)
    tk, tp, ti = csr(d.ui_train)
    sk, sp, si = csr(d.ui_test)
    if social:
        fk, fp, fi = csr(d.user_friends)
        extra.update(f_keys=fk, f_ptr=fp, f_items=fi)
    return {**extra, "csv_u": np.array([r[0] for r in rows], dtype=np.int64), "csv_i": np.array([r[1] for r in rows], dtype=np.int64),
            "seed": seed, "user_nums": d.user_nums, "item_nums": d.item_nums, "train_keys": tk, "train_ptr": tp,
            "train_items": ti, "test_keys": sk, "test_ptr": sp, "test_items": si}


def ml100k_anchor(ref, sampler):
    """Hashes only (MovieLens data may not be redistributed): seed(0) -> RankingPreprocess(ml-100k
    defaults) -> seed(1) -> samplers.  Re-checkable wherever the dataset file exists."""
    path = os.path.join(ref, "dataset", "ml-100k", "u.data")
    if not os.path.exists(path):
        return None
    from model.RankingPreprocess import RankingPreprocess
    cfg = {"data.root_dir": os.path.join(ref, "dataset"), "data.dataset": "ml-100k", "data.file_name": "u.data",
           "data.sep": "\t", "data.format": "UI", "data.split_way": "rs", "data.split_ratio": "[0.7,0.2,0.1]",
           "data.split_by_time": "False", "data.user_min": "0", "data.item_min": "5", "test.neg_samples": "0",
           "recommender": "BPR"}
    np.random.seed(0)
    d = RankingPreprocess(cfg, logging.getLogger("golden"))

    def h(a):
        return hashlib.sha256(np.ascontiguousarray(a).astype("<i8").tobytes()).hexdigest()[:16]

    out = {"user_nums": d.user_nums, "item_nums": d.item_nums, "n_train": sum(len(v) for v in d.ui_train.values()),
           "n_test": sum(len(v) for v in d.ui_test.values()), "n_test_users": len(d.ui_test)}
    tk, tp, ti = csr(d.ui_train)
    out["train_hash"] = [h(tk), h(tp), h(ti)]
    np.random.seed(1)
    nb, u, i, j = sampler.pairwise_ranking_sampler(d, 4, 6144)
    out["pairwise"] = {"batches": nb, "n": len(u), "first": [[int(u[k]), int(i[k]), int(j[k])] for k in range(4)],
                       "hash": [h(u), h(i), h(j)], "next": int(np.random.randint(2 ** 30))}
    np.random.seed(1)
    nb, u, i, y = sampler.pointwise_ranking_sampler(d, 4, 6144)
    out["pointwise"] = {"batches": nb, "n": len(u), "hash": [h(u), h(i)], "ysum": float(y.sum())}
    np.random.seed(1)
    nb, u, i, neg = sampler.ranking_sampler_cml(d, 20, 6144)
    out["cml"] = {"batches": nb, "n": len(u), "hash": [h(u), h(i), h(neg)]}
    return out


def rating_case(ref):
    """RatingPreprocess._read_file / _load_data on a small libfm-ish pair of files (the FM input contract)."""
    from model.RatingPreprocess import RatingPreprocess
    g = np.random.Generator(np.random.PCG64(21))
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "syn"))
    lines = {}
    for name, n in (("syn.train.libfm", 300), ("syn.test.libfm", 80)):
        rows = ["%d,%d:1,%d:1" % (g.integers(1, 6), g.integers(1, 60), 60 + g.integers(1, 40)) for _ in range(n)]
        rows[3] = "4,7:0.5,%d:2" % (60 + 5)  # real-valued features
        lines[name] = rows
        with open(os.path.join(tmp, "syn", name), "w") as f:
            f.write("\n".join(rows) + "\n")
    cfg = {"recommender": "FM", "data.root_dir": tmp, "data.dataset": "syn", "train": ".train.libfm", "test": ".test.libfm",
           "is_real_valued": "True"}
    d = RatingPreprocess(cfg, logging.getLogger("golden"))
    out = {"train_lines": np.array(lines["syn.train.libfm"]), "test_lines": np.array(lines["syn.test.libfm"]),
           "feature_nums": d.feature_nums, "X_idx_tr": np.array(d.X_idx_tr, dtype=np.int64),
           "X_value_tr": np.array(d.X_value_tr, dtype=np.float64), "y_tr": np.array(d.y_tr, dtype=np.float64),
           "X_idx_t": np.array(d.X_idx_t, dtype=np.int64), "X_value_t": np.array(d.X_value_t, dtype=np.float64),
           "y_t": np.array(d.y_t, dtype=np.float64)}
    # anchor on the one real FM file in the checkout (hashes only)
    real = os.path.join(ref, "dataset", "ml-1m", "ml-1m.test.libfm")
    if os.path.exists(real):
        o = RatingPreprocess.__new__(RatingPreprocess)
        o.is_real_valued, o.all_features = True, {}
        xi, xv, yy = o._read_file(real)
        idx = np.array([[o.all_features[c] for c in r] for r in xi], dtype=np.int64)
        h = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
        out.update(ml1m_rows=len(yy), ml1m_features=len(o.all_features), ml1m_idx_hash=h(idx),
                   ml1m_val_hash=h(np.array(xv, dtype=np.float64)), ml1m_y_hash=h(np.array(yy, dtype=np.float64)))
    return out


def libfm_case(ref):
    """utils/fm_to_libfm.py run as the script it is, inside a scratch tree: cwd = <tmp>/utils so that its
    '../dataset/ml-1m/ratings.dat' is a synthetic file and its outputs land in <tmp>.  The golden holds the input
    text, the seed and the two output texts."""
    import runpy
    g = np.random.Generator(np.random.PCG64(33))
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "utils"))
    os.makedirs(os.path.join(tmp, "dataset", "ml-1m"))
    users = g.choice(np.arange(1, 400), size=90, replace=False)
    items = g.choice(np.arange(1, 4000), size=250, replace=False)
    lines = ["%d::%d::%d::%d" % (g.choice(users), g.choice(items), g.integers(1, 6), 956703932 + g.integers(0, 10 ** 7))
             for _ in range(1500)]
    text = "\n".join(lines) + "\n"
    with open(os.path.join(tmp, "dataset", "ml-1m", "ratings.dat"), "w") as f:
        f.write(text)
    cwd = os.getcwd()
    sys.path.insert(0, os.path.join(ref, "utils"))
    os.chdir(os.path.join(tmp, "utils"))
    try:
        np.random.seed(7)
        runpy.run_path(os.path.join(ref, "utils", "fm_to_libfm.py"), run_name="__main__")
    finally:
        os.chdir(cwd)
        sys.path.pop(0)
    out = {"seed": 7, "ratings": text}
    for part in ("train", "test"):
        with open(os.path.join(tmp, "dataset", "ml-1m", "ml-1m.%s.libfm" % part)) as f:
            out[part] = f.read()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    sampler, metrics = _import_reference(a.ref)

    toy = {0: [1, 2, 3], 1: [4, 5], 2: [0, 9]}
    cases = [
        sampler_case(sampler, "toy", toy, 10, 123, 2, 4),
        sampler_case(sampler, "syn200", synth_ui(1, 200, 300, 3, 40, skip_every=17), 300, 5, 4, 512),
        # dense users: heavy seen-set rejection and frequent in-group duplicates
        sampler_case(sampler, "dense", synth_ui(2, 24, 16, 8, 11), 16, 2 ** 32 - 1, 4, 64),
        # item_nums a power of two (mask == max: no masked rejections) and neg_ratio 20 (CML conf)
        sampler_case(sampler, "pow2", synth_ui(3, 60, 64, 2, 12), 64, 0, 20, 256),
    ]
    for name, out in cases:
        np.savez_compressed(os.path.join(a.out, "sampler_%s.npz" % name), **out)
    np.savez_compressed(os.path.join(a.out, "sampler_sbpr.npz"), **sbpr_case(sampler))
    np.savez_compressed(os.path.join(a.out, "sampler_sbpr_tiny.npz"), **sbpr_case(sampler, seed=11, n_users=9, n_items=40, neg_ratio=2, batch=8))
    with open(os.path.join(a.out, "metrics.json"), "w") as f:
        json.dump(metrics_cases(metrics), f)
    np.savez_compressed(os.path.join(a.out, "sp_mat.npz"), **sp_mat_case())
    np.savez_compressed(os.path.join(a.out, "cune.npz"), **cune_case())
    np.savez_compressed(os.path.join(a.out, "preprocess_syn.npz"), **preprocess_case(a.ref))
    np.savez_compressed(os.path.join(a.out, "preprocess_social.npz"), **preprocess_case(a.ref, social=True))
    np.savez_compressed(os.path.join(a.out, "rating_syn.npz"), **rating_case(a.ref))
    with open(os.path.join(a.out, "libfm_syn.json"), "w") as f:
        json.dump(libfm_case(a.ref), f)
    anchor = ml100k_anchor(a.ref, sampler)
    if anchor is not None:
        with open(os.path.join(a.out, "ml100k_anchor.json"), "w") as f:
            json.dump(anchor, f, indent=1)
    print("golden vectors written to", os.path.abspath(a.out))


if __name__ == "__main__":
    main()
