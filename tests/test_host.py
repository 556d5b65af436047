"""CPU-side checks of the product: the C-ABI library loads and exports every symbol of
include/crec.h, the host RNG + samplers are bit-exact against the reference's golden vectors
and numpy, and device entry points fail loudly without a GPU.  No compute on a GPU here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from cleverrec_amd import _lib, build as crec_build
from cleverrec_amd.data import RankingData, csr_from_dict
from cleverrec_amd.utils import sampler as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    crec_build.build()
    c = _lib.Context(None)
    yield c
    c.close()


def test_library_exports_every_declared_symbol():
    crec_build.build()
    hdr = open(os.path.join(ROOT, "include", "crec.h")).read()
    declared = set(re.findall(r"\b(crec_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    lib = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "libcrec.so does not export %s" % name
    assert declared == set(_lib.exported_symbols()), "ctypes table out of step with crec.h"
    assert _lib.load().crec_version() >= 100


def test_rng_matches_numpy(ctx):
    for seed in (0, 123, 2 ** 32 - 1):
        ctx.seed(seed)
        np.random.seed(seed)
        for n in [1, 2, 3, 10, 1349, 3706, 4096, 4097, 2 ** 31, 2 ** 32, 1, 7] * 30:
            assert ctx.randint(n) == int(np.random.randint(n))
        for n in (0, 1, 2, 63, 64, 65, 1000, 5000):
            assert np.array_equal(ctx.permutation(n), np.random.permutation(n))


@pytest.mark.parametrize("name", ["toy", "syn200", "dense", "pow2"])
def test_samplers_bit_exact_vs_reference(ctx, golden_dir, name):
    g = np.load(os.path.join(golden_dir, "sampler_%s.npz" % name))
    csr = (g["keys"], g["ptr"], g["items"])
    item_nums, seed, nr = int(g["item_nums"]), int(g["seed"]), int(g["neg_ratio"])
    ctx.seed(seed)
    u, i, j = ctx.sample_pairwise(csr, item_nums, nr)
    assert np.array_equal(u, g["pw_u"]) and np.array_equal(i, g["pw_i"]) and np.array_equal(j, g["pw_j"])
    assert ctx.randint(2 ** 30) == int(g["pw_next"])
    ctx.seed(seed)
    u, i, y = ctx.sample_pointwise(csr, item_nums, nr)
    assert np.array_equal(u, g["pt_u"]) and np.array_equal(i, g["pt_i"]) and np.array_equal(y, g["pt_y"])
    assert ctx.randint(2 ** 30) == int(g["pt_next"])
    ctx.seed(seed)
    u, i, neg = ctx.sample_cml(csr, item_nums, nr)
    assert np.array_equal(u, g["cml_u"]) and np.array_equal(i, g["cml_i"]) and np.array_equal(neg, g["cml_neg"])
    assert ctx.randint(2 ** 30) == int(g["cml_next"])


def test_sampler_wrappers_keep_numpy_global_stream_in_step(golden_dir):
    """Drop-in behaviour: np.random.seed(s); sampler(...) leaves np.random exactly where the
    reference's sampler would (pw_next was drawn from np.random after the reference call)."""
    g = np.load(os.path.join(golden_dir, "sampler_syn200.npz"))

    class D:
        pass
    d = D()
    d.ui_train = {int(k): g["items"][g["ptr"][n]:g["ptr"][n + 1]].tolist() for n, k in enumerate(g["keys"])}
    d.item_nums = int(g["item_nums"])
    np.random.seed(int(g["seed"]))
    nb, u, i, j = S.pairwise_ranking_sampler(d, int(g["neg_ratio"]), int(g["batch"]))
    assert nb == int(g["pw_batches"]) and np.array_equal(j, g["pw_j"])
    assert int(np.random.randint(2 ** 30)) == int(g["pw_next"])
    np.random.seed(int(g["seed"]))
    nb, u, i, y = S.pointwise_ranking_sampler(d, int(g["neg_ratio"]), int(g["batch"]))
    assert nb == int(g["pt_batches"]) and np.array_equal(i, g["pt_i"])
    assert int(np.random.randint(2 ** 30)) == int(g["pt_next"])
    np.random.seed(int(g["seed"]))
    nb, u, i, neg = S.ranking_sampler_cml(d, int(g["neg_ratio"]), int(g["batch"]))
    assert nb == int(g["cml_batches"]) and np.array_equal(neg, g["cml_neg"])
    assert int(np.random.randint(2 ** 30)) == int(g["cml_next"])


def test_sampler_edge_cases(ctx):
    # empty ui_train
    e = (np.zeros(0, np.int32), np.zeros(1, np.int64), np.zeros(0, np.int32))
    u, i, j = ctx.sample_pairwise(e, 10, 4)
    assert len(u) == 0
    # a user without rows between users with rows, and neg_ratio 0
    csr = csr_from_dict({0: [1], 1: [], 2: [0, 3]})
    u, i, j = ctx.sample_pairwise(csr, 5, 0)
    assert len(u) == 0
    # a user who cannot supply neg_ratio distinct unseen items: the reference would spin forever
    with pytest.raises(_lib.CrecError):
        ctx.sample_pairwise(csr_from_dict({0: [0, 1, 2]}), 4, 2)
    # item id out of range
    with pytest.raises(_lib.CrecError):
        ctx.sample_pairwise(csr_from_dict({0: [7]}), 4, 1)
    # item_nums == 1 and the only item unseen is impossible; item_nums 2 with one seen: always the other
    ctx.seed(3)
    u, i, j = ctx.sample_pairwise(csr_from_dict({5: [1]}), 2, 1)
    assert u.tolist() == [5] and i.tolist() == [1] and j.tolist() == [0]


def test_device_entry_points_fail_loudly_without_gpu(ctx):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    t = _lib.BprTables()
    a = _lib.Adam()
    rc = ctx.lib.crec_bpr_train_epoch(ctx.h, C.byref(t), C.byref(a), None, None, None, 0, 1, None, None)
    assert rc == _lib.EINVAL and b"host-only" in ctx.lib.crec_last_error(ctx.h)
    with pytest.raises(_lib.CrecError):
        _lib.Context(0)  # no device: ENODEVICE, never a silent CPU path
    from cleverrec_amd.runtime import Session
    with pytest.raises(RuntimeError):
        Session(0)


def test_ranking_data_protocol():
    d = RankingData(5, 6, {0: [3, 1], 2: [5], 4: [0, 2, 4]}, {0: [2], 4: [1, 5]})
    assert d.test_users.tolist() == [0, 4] and d.n_train == 6
    assert d.seen_csr[0].tolist() == [0, 2, 2, 3, 3, 6] and d.seen_csr[1].tolist() == [1, 3, 5, 0, 2, 4]
    assert d.test_csr[0].tolist() == [0, 1, 1, 1, 1, 3] and d.test_csr[1].tolist() == [2, 1, 5]
    assert d.ui_train == {0: [3, 1], 2: [5], 4: [0, 2, 4]}


def test_threaded_sampler_path_is_bit_identical(tmp_path):
    """Epochs of >= 65536 draws take the pipelined path (RNG thread, swap thread, gather workers): same arrays and
    same RNG end state as the serial path (CREC_SAMPLER_THREADS=0), and as the independent C oracle."""
    import subprocess
    import sys
    script = r'''
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
from cleverrec_amd import _lib
from cleverrec_amd.data import synthetic
data = synthetic("S-ml100k")
ctx = _lib.Context(-1)
h = hashlib.sha256()
for fn, nr in ((ctx.sample_pairwise, 4), (ctx.sample_pointwise, 4), (ctx.sample_cml, 20)):
    ctx.seed(11)
    for a in fn(data.train_csr, data.item_nums, nr):
        h.update(np.ascontiguousarray(a).tobytes())
    h.update(str(ctx.randint(2 ** 30)).encode())
print(h.hexdigest())
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for threads in ("0", "1", "3"):
        env = dict(os.environ, CREC_SAMPLER_THREADS=threads)
        out.append(subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, check=True).stdout.strip())
    assert out[0] == out[1] == out[2] and len(out[0]) == 64
    # and the serial path against the oracle's own sampler on the same data
    import oracle
    from cleverrec_amd.data import synthetic
    data = synthetic("S-ml100k")
    c = _lib.Context(-1)
    c.seed(11)
    u, i, j = c.sample_pairwise(data.train_csr, data.item_nums, 4)
    nb, ou, oi, oj = oracle.sample_pairwise(oracle.Rng(11), data.ui_train, data.item_nums, 4, 6144)
    assert len(u) >= 65536 and np.array_equal(u, ou) and np.array_equal(i, oi) and np.array_equal(j, oj)


def test_grid_search_order_and_types():
    """cleverrec_amd.config.grid: the reference's loop nest (main_tuning.py:38-47): embed_size outermost, neg_ratio
    innermost, ints / floats as it types them; scalars count as one-element lists."""
    from cleverrec_amd.config import grid
    pts = list(grid({"embed_size": "[8,16]", "reg": "[0.1,0.01]", "neg_ratio": "[2,4]"}))
    assert len(pts) == 8
    assert pts[0] == {"embed_size": 8, "reg": 0.1, "neg_ratio": 2} and pts[1] == {"embed_size": 8, "reg": 0.1, "neg_ratio": 4}
    assert pts[2]["reg"] == 0.01 and pts[4]["embed_size"] == 16
    assert all(isinstance(p["embed_size"], int) and isinstance(p["reg"], float) and isinstance(p["neg_ratio"], int) for p in pts)
    assert list(grid({"embed_size": "64", "reg": "1e-2", "neg_ratio": " [4] "})) == [{"embed_size": 64, "reg": 0.01, "neg_ratio": 4}]


def test_sampler_prefetch_is_stream_exact(golden_dir):
    """utils/sampler.py prefetch: epochs drawn ahead on a host thread give the same arrays and leave np.random in the
    same state as the serial order -- against the reference-generated golden stream for the first call, against the
    serial path for the following epochs; a foreign np.random draw between epochs drops the prefetch (miss) and the
    results still equal the serial order."""
    g = np.load(os.path.join(golden_dir, "sampler_syn200.npz"))

    class D:
        pass
    data = D()
    data.ui_train = {int(k): g["items"][g["ptr"][n]:g["ptr"][n + 1]].tolist() for n, k in enumerate(g["keys"])}
    data.item_nums = int(g["item_nums"])
    seed, nr, B = int(g["seed"]), int(g["neg_ratio"]), int(g["batch"])

    def run(prefetch, disturb):
        S.drop_prefetch()
        np.random.seed(seed)
        outs = []
        for e in range(4):
            outs.append(S.pairwise_ranking_sampler(data, nr, B, prefetch_next=prefetch))
            if disturb and e == 1:
                outs.append(np.random.randint(1000))  # somebody else draws between two epochs
        outs.append(np.random.randint(2 ** 30))  # where the stream stands afterwards
        return outs

    def same(a, b):
        return all((x == y) if np.isscalar(x) else all(np.array_equal(p, q) for p, q in zip(x, y)) for x, y in zip(a, b))
    base = run(False, False)
    assert np.array_equal(base[0][1], g["pw_u"]) and np.array_equal(base[0][3], g["pw_j"])  # the reference's own stream
    h0 = dict(S.stats)
    assert same(run(True, False), base)
    assert S.stats["prefetch_hits"] - h0["prefetch_hits"] == 3 and S.stats["prefetch_misses"] == h0["prefetch_misses"]
    dbase = run(False, True)
    assert same(run(True, True), dbase) and S.stats["prefetch_misses"] - h0["prefetch_misses"] == 1
    S.drop_prefetch()


def test_properties_outside_the_built_shapes_are_refused_up_front():
    """batch_size, topk and embed_size beyond include/crec.h's shapes raise a ValueError naming key, value and limit when
    the model object is built -- not a CREC_EINVAL in the middle of an epoch (the reference has no such limits)."""
    from cleverrec_amd import config
    base = {"recommender": "BPR", "batch_size": "6144", "topk": "[10,20]", "embed_size": "64"}
    config.validate(base)
    config.validate(dict(base, batch_size="10000"))  # BPR indexes batches beyond 8192 in global memory (csrc/index_big.hip)
    with pytest.raises(ValueError) as e:            # the other models' steps sort a batch in LDS
        config.validate(dict(base, recommender="CML", batch_size="100000"))
    assert "batch_size = 100000" in str(e.value)
    config.validate(dict(base, recommender="CML", batch_size="10000"))
    config.validate(dict(base, recommender="NeuMF", batch_size="10000"))  # NCF / FM index a batch's slots in several chunks
    for key, val, frag in (("batch_size", "600000", "batch_size = 600000"), ("topk", "[10,20,200]", "topk = [10,20,200]"),
                           ("embed_size", "300", "embed_size = 300"), ("topk", "[20,10]", "ascending")):
        with pytest.raises(ValueError) as e:
            config.validate(dict(base, **{key: val}))
        assert frag in str(e.value)
    config.validate(dict(base, topk="[10,50,100]"))  # lists beyond 32 are ranked in passes of 32 (crec_eval_topk)
    config.validate(dict(base, embed_size="30"))     # BPR pads its rows to whole float4 slices itself
    with pytest.raises(ValueError) as e:             # the other graphs couple the columns: their width is taken as given
        config.validate(dict(base, recommender="CML", embed_size="30"))
    assert "embed_size = 30" in str(e.value)
    config.validate(dict(base, embed_size="252"), extra_width=4)
    with pytest.raises(ValueError):
        config.validate(dict(base, embed_size="256"), extra_width=4)  # CML widens its rows by 4 columns for evaluation
    config.validate(dict(base, embed_size="[32,64]"))  # a grid list (main_tuning.py) is validated per run


def test_initializers_have_the_reference_distributions():
    """get_initializer (utils/tools.py:51-63) -> the TF1 initializers' distributions: normal N(0, s^2); tnormal = the parent
    normal N(0, s^2) cut at 2 s (std 0.8796 s); uniform U(-s, s); xavier = Glorot-uniform, limit sqrt(6 / (fan_in + fan_out));
    xavier_normal = truncated normal of parent std sqrt(1.3 * 2 / (fan_in + fan_out)) (tf.contrib's variance_scaling, FAN_AVG).
    TF's Philox stream itself is not reproducible (and the reference never seeds it): moments and supports are what can match."""
    import math
    import torch
    from cleverrec_amd.utils.tools import get_initializer
    g = lambda: torch.Generator().manual_seed(7)
    n, d, s = 4000, 64, 0.05
    x = get_initializer("normal", s, g())([n, d]).double()
    assert abs(float(x.mean())) < 4 * s / math.sqrt(n * d) and abs(float(x.std()) / s - 1) < 0.01
    x = get_initializer("tnormal", s, g())([n, d]).double()
    assert float(x.abs().max()) <= 2 * s and abs(float(x.std()) / (0.8796 * s) - 1) < 0.01
    x = get_initializer("uniform", s, g())([n, d]).double()
    assert float(x.abs().max()) <= s and abs(float(x.std()) / (s / math.sqrt(3)) - 1) < 0.01
    lim = math.sqrt(6.0 / (n + d))
    for name in ("xavier", "xavier_uniform"):  # the shipped NCF confs say xavier_uniform, which the reference does not handle
        x = get_initializer(name, s, g())([n, d]).double()
        assert float(x.abs().max()) <= lim and abs(float(x.std()) / (lim / math.sqrt(3)) - 1) < 0.01
    sd = math.sqrt(1.3 * 2.0 / (n + d))
    x = get_initializer("xavier_normal", s, g())([n, d]).double()
    assert float(x.abs().max()) <= 2 * sd and abs(float(x.std()) / (0.8796 * sd) - 1) < 0.01
    # a vector (biases, h) takes fan_in = fan_out = its length; the same generator state gives the same tensor
    x = get_initializer("xavier", s, g())([d]).double()
    assert float(x.abs().max()) <= math.sqrt(6.0 / (2 * d))
    assert torch.equal(get_initializer("normal", s, g())([8, 4]), get_initializer("normal", s, g())([8, 4]))
    assert get_initializer("normal", s)([3, 5]).dtype == torch.float32
    with pytest.raises(ValueError):
        get_initializer("he_uniform", s)([2, 2])  # conf/SBPR.properties mentions it in a comment; the reference returns None and fails


@pytest.mark.parametrize("name", ["sampler_sbpr", "sampler_sbpr_tiny"])
def test_sbpr_sampler_matches_reference_stream(golden_dir, name):
    """get_SPu + ranking_sampler_sbpr (utils/tools.py:116-127, utils/sampler.py:102-141) against vectors generated by the
    reference itself: the SPu lists in the reference's list order, the five shuffled columns, and where np.random stands."""
    from cleverrec_amd.utils.tools import get_SPu
    g = np.load(os.path.join(golden_dir, name + ".npz"))

    class D:
        pass
    d = D()
    rows = lambda k, p, v: {int(key): v[p[n]:p[n + 1]].tolist() for n, key in enumerate(k)}
    d.ui_train = rows(g["keys"], g["ptr"], g["items"])
    d.user_friends = rows(g["f_keys"], g["f_ptr"], g["f_items"])
    d.item_nums = int(g["item_nums"])
    SPu = get_SPu(d)
    assert SPu == rows(g["spu_keys"], g["spu_ptr"], g["spu_items"])  # same users, same LIST order
    assert list(SPu.keys()) == [int(k) for k in g["spu_keys"]]
    np.random.seed(int(g["seed"]))
    nb, u, i, i_s, i_neg, suk = S.ranking_sampler_sbpr(d, SPu, int(g["neg_ratio"]), int(g["batch"]))
    assert nb == int(g["batches"])
    for got, want in ((u, "u"), (i, "i"), (i_s, "i_s"), (i_neg, "i_neg"), (suk, "suk")):
        assert np.array_equal(np.asarray(got).astype(np.int64), g[want]), want
    assert int(np.random.randint(2 ** 30)) == int(g["next"])


def test_get_sp_mat_matches_reference_lists():
    """utils/tools.py:100-113 (TransCF): the CSRs of cleverrec_amd.utils.tools.get_sp_mat against the index / value lists the
    reference's own loops hand to tf.SparseTensor (tests/golden/sp_mat.npz, tools/make_golden.py:sp_mat_case): the same
    entries, per row in the same order, and weights float32(1 / row length)."""
    import os
    from cleverrec_amd.utils.tools import get_sp_mat
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "sp_mat.npz"))
    keys, ptr, items = z["keys"], z["ptr"], z["items"]

    class D:
        pass
    D.user_nums, D.item_nums = int(z["user_nums"]), int(z["item_nums"])
    D.ui_train = {int(k): [int(x) for x in items[ptr[n]:ptr[n + 1]]] for n, k in enumerate(keys)}
    ui_ptr, ui_items, iu_ptr, iu_users = get_sp_mat(D)
    assert list(z["ui_shape"]) == [D.user_nums, D.item_nums] and list(z["iu_shape"]) == [D.item_nums, D.user_nums]
    for (idx, val, rptr, cols, n_rows) in ((z["ui_indices"], z["ui_values"], ui_ptr, ui_items, D.user_nums),
                                            (z["iu_indices"], z["iu_values"], iu_ptr, iu_users, D.item_nums)):
        assert len(rptr) == n_rows + 1 and rptr[-1] == len(idx)
        for r in range(n_rows):
            sel = idx[:, 0] == r  # the reference's entries of row r, in list order
            np.testing.assert_array_equal(cols[rptr[r]:rptr[r + 1]], idx[sel, 1])
            n = int(rptr[r + 1] - rptr[r])
            if n:  # the device forms float32(1.0 / n) (csrc/transcf.hip nbr_weight); TF casts the list's float64 the same way
                np.testing.assert_array_equal(val[sel].astype(np.float32), np.full(n, np.float32(1.0 / n)))


def _cune_toy(seed=3, U=40, I=30):
    from cleverrec_amd.data import RankingData
    g = np.random.Generator(np.random.PCG64(seed))
    ui_train = {u: [int(x) for x in g.choice(I, size=int(g.integers(1, 7)), replace=False)] for u in range(U) if u % 11 != 10}
    ui_train[U - 1] = [I - 1]  # a user sharing nothing with anybody stays outside the network (item I-1 removed from the others)
    for u in ui_train:
        if u != U - 1:
            ui_train[u] = [x for x in ui_train[u] if x != I - 1] or [0]
    return RankingData(U, I, ui_train, {0: [1]})


def test_cunet_is_the_pairwise_intersection_network():
    """utils/tools.py:132-141 by its definition (U^2 set intersections) against the sparse product of utils/tools.cunet."""
    from cleverrec_amd.utils.tools import cunet
    data = _cune_toy()
    nbr_weight, draw_list = cunet(data)
    exp_w, exp_l = {}, {}
    for u1 in data.ui_train:
        for u2 in data.ui_train:
            w = len(set(data.ui_train[u1]) & set(data.ui_train[u2])) if u1 != u2 else 0
            if w > 0:
                exp_w.setdefault(u1, {})[u2] = w
                exp_l.setdefault(u1, []).extend([u2] * w)
    assert list(nbr_weight) == list(exp_w) and max(data.ui_train) not in nbr_weight
    for u in exp_w:
        assert list(nbr_weight[u].items()) == list(exp_w[u].items())  # same neighbours, weights AND order
        assert draw_list[u].tolist() == exp_l[u]


def test_deep_walks_follow_the_network_and_the_numpy_stream():
    from cleverrec_amd.utils.tools import cunet, deep_walks
    data = _cune_toy()
    nbr_weight, draw_list = cunet(data)
    np.random.seed(5)
    walks = deep_walks(nbr_weight, draw_list, walk_count=4, walk_length=7)
    after = np.random.randint(1 << 30)
    np.random.seed(5)
    again = deep_walks(nbr_weight, draw_list, 4, 7)
    assert walks == again and np.random.randint(1 << 30) == after  # a function of the np.random state, which it advances
    assert len(walks) == 4 * len(draw_list) and all(len(w) == 7 for w in walks)
    starts = sorted(int(w[0]) for w in walks)
    assert starts == sorted(list(draw_list) * 4)
    for w in walks:
        for a, b in zip(w, w[1:]):
            assert int(b) in nbr_weight[int(a)]
    # the first step of a user's first walk: its heaviest neighbour, the first of equals in set order
    first = {}
    np.random.seed(5)
    for w in deep_walks(nbr_weight, draw_list, 1, 2)[:]:
        first[int(w[0])] = int(w[1])
    u0 = next(iter(draw_list))
    assert nbr_weight[u0][first[u0]] == max(nbr_weight[u0].values())


def test_topk_friends_and_spu():
    """get_topK_friends_and_SPu (utils/tools.py:130-207) end to end on a toy set: friends by cosine similarity of the walk
    vectors (checked against the vectors the function saw), SPu by the set arithmetic of get_SPu."""
    from cleverrec_amd.utils import tools as T
    data = _cune_toy()
    seen = {}
    orig = T.walk_embeddings

    def spy(walks, size, window, **kw):
        seen["vec"] = orig(walks, size, window, **kw)
        return seen["vec"]
    T.walk_embeddings = spy
    try:
        np.random.seed(2)
        friends, SPu = T.get_topK_friends_and_SPu(data, 5, 8, 12, 3, 3)
    finally:
        T.walk_embeddings = orig
    vec = seen["vec"]
    assert all(v.shape == (12,) and np.isfinite(v).all() for v in vec.values())
    users = list(friends)
    for u in users[:10]:
        sims = {v: float(vec[str(u)] @ vec[str(v)] / (np.linalg.norm(vec[str(u)]) * np.linalg.norm(vec[str(v)]))) for v in users if v != u}
        best = sorted(sims.values(), reverse=True)[:3]
        assert len(friends[u]) == 3 and u not in friends[u]
        np.testing.assert_allclose([sims[f] for f in friends[u]], best, rtol=1e-5)
    assert SPu == T.get_SPu(data, friends)
    # users that walk together end up closer than users that never meet: the vectors carry the network
    nbr_weight, _ = T.cunet(data)
    close, far = [], []
    for u in users:
        for v in users:
            if u < v:
                c = float(vec[str(u)] @ vec[str(v)] / (np.linalg.norm(vec[str(u)]) * np.linalg.norm(vec[str(v)])))
                (close if v in nbr_weight[u] and nbr_weight[u][v] >= 2 else far).append(c)
    assert np.mean(close) > np.mean(far)


def test_cune_walks_friends_and_spu_match_the_reference_run(golden_dir):
    """tests/golden/cune.npz: the reference's own get_topK_friends_and_SPu (utils/tools.py:130-207) run by tools/make_golden.py
    with a recorder in gensim's place: the deep walks (np.random stream, set-iteration tie-breaks, shuffle) bit for bit, the
    stream position afterwards, and -- given the same stand-in vectors -- the top-K friends in order and the SPu lists in order."""
    from cleverrec_amd.data import RankingData
    from cleverrec_amd.utils import tools as T
    z = np.load(os.path.join(golden_dir, "cune.npz"))
    ui_train = {int(k): [int(x) for x in z["items"][z["ptr"][n]:z["ptr"][n + 1]]] for n, k in enumerate(z["keys"])}
    data = RankingData(int(z["user_nums"]), int(z["item_nums"]), ui_train, {})
    wc, wl, dim, win, topk = (int(x) for x in z["params"])
    assert list(z["w2v_args"]) == [dim, win, 0, 3]  # size, window, min_count, iter as the reference passes them (:173)
    got = {}

    def vectors(walks, size, window, **kw):  # tools/make_golden.py: cune_vectors
        got["walks"] = walks
        return {t: np.random.Generator(np.random.PCG64([77, int(t)])).standard_normal(size).astype(np.float32) for w in walks for t in w}
    orig, T.walk_embeddings = T.walk_embeddings, vectors
    try:
        np.random.seed(int(z["seed"]))
        friends, SPu = T.get_topK_friends_and_SPu(data, wc, wl, dim, win, topk)
        nxt = np.random.randint(2 ** 30)
    finally:
        T.walk_embeddings = orig
    assert np.array_equal(np.array([[int(t) for t in w] for w in got["walks"]]), z["walks"])
    assert nxt == int(z["next"])
    ref_friends = {int(k): [int(x) for x in z["f_items"][z["f_ptr"][n]:z["f_ptr"][n + 1]]] for n, k in enumerate(z["f_keys"])}
    ref_spu = {int(k): [int(x) for x in z["spu_items"][z["spu_ptr"][n]:z["spu_ptr"][n + 1]]] for n, k in enumerate(z["spu_keys"])}
    assert list(friends) == list(ref_friends) and all(list(friends[u]) == ref_friends[u] for u in friends)
    assert list(SPu) == list(ref_spu) and all(list(SPu[u]) == ref_spu[u] for u in SPu)


def test_sampler_fast_division_is_exact():
    """The gather of every epoch sample divides the shuffled index by neg_ratio with a multiply and two shifts
    (crec_sampler.cpp: FastDiv); against integer division for every divisor up to 300 and a few large ones, on the edges
    of the 32-bit range and on random dividends."""
    import ctypes as C
    from cleverrec_amd import _lib
    lib = _lib.load()
    g = np.random.Generator(np.random.PCG64(3))
    ns = np.concatenate([np.arange(0, 2000), 2 ** 32 - 1 - np.arange(0, 2000), g.integers(0, 2 ** 32, 20000),
                         (2 ** np.arange(1, 32))[:, None].repeat(3, 1).ravel() + np.tile([-1, 0, 1], 31)]).astype(np.uint32)
    out = np.empty_like(ns)
    for d in list(range(1, 301)) + [1000, 4097, 65535, 65536, 65537, 2 ** 31 - 1, 2 ** 31, 2 ** 31 + 1, 2 ** 32 - 1]:
        top = (2 ** 32 - 1) // d
        mult = np.array([max(top - 2, 0) * d, max(top - 1, 0) * d, top * d], dtype=np.uint64)  # the largest multiples of d and their neighbours
        extra = np.concatenate([np.maximum(mult, 1) - 1, mult, np.minimum(mult + 1, 2 ** 32 - 1)]).astype(np.uint32)
        for arr in (ns, extra):
            o = out[:len(arr)]
            rc = lib.crec_debug_fastdiv(C.c_uint32(d), arr.ctypes.data_as(C.c_void_p), C.c_int64(len(arr)), o.ctypes.data_as(C.c_void_p))
            assert rc == 0
            assert np.array_equal(o, (arr.astype(np.uint64) // d).astype(np.uint32)), d


def test_sbpr_sampler_prefetch_is_stream_exact(golden_dir):
    """The SBPR / CUNE_BPR sampler drawn ahead on a host thread: the same five columns and the same np.random state as the
    serial order (first epoch = the reference's own stream, tests/golden/sampler_sbpr.npz), hits when nobody draws between
    epochs, a miss (and still the serial results) when somebody does or when another SPu dict is passed."""
    from cleverrec_amd.utils.tools import get_SPu
    g = np.load(os.path.join(golden_dir, "sampler_sbpr.npz"))

    class D:
        pass
    d = D()
    rows = lambda k, p, v: {int(key): v[p[n]:p[n + 1]].tolist() for n, key in enumerate(k)}
    d.ui_train, d.user_friends, d.item_nums = rows(g["keys"], g["ptr"], g["items"]), rows(g["f_keys"], g["f_ptr"], g["f_items"]), int(g["item_nums"])
    SPu = get_SPu(d)
    seed, nr, B = int(g["seed"]), int(g["neg_ratio"]), int(g["batch"])

    def run(prefetch, disturb):
        S.drop_prefetch()
        np.random.seed(seed)
        outs = []
        for e in range(4):
            outs.append(S.ranking_sampler_sbpr(d, SPu, nr, B, is_suk=(e % 2 == 0), prefetch_next=prefetch))
            if disturb and e == 1:
                outs.append(np.random.randint(1000))
        outs.append(np.random.randint(2 ** 30))
        return outs

    def same(a, b):
        return all((x == y) if np.isscalar(x) else all(np.array_equal(p, q) for p, q in zip(x, y)) for x, y in zip(a, b))
    base = run(False, False)
    assert np.array_equal(np.asarray(base[0][1]).astype(np.int64), g["u"]) and np.array_equal(np.asarray(base[0][4]).astype(np.int64), g["i_neg"])
    assert len(base[0]) == 6 and len(base[1]) == 5  # is_suk=False drops the suk column, not the draws
    h0 = dict(S.stats)
    assert same(run(True, False), base)
    assert S.stats["prefetch_hits"] - h0["prefetch_hits"] == 3 and S.stats["prefetch_misses"] == h0["prefetch_misses"]
    dbase = run(False, True)
    assert same(run(True, True), dbase) and S.stats["prefetch_misses"] - h0["prefetch_misses"] == 1
    # another SPu dict (same content, another object): the pending sample is not trusted
    S.drop_prefetch()
    np.random.seed(seed)
    first = S.ranking_sampler_sbpr(d, SPu, nr, B, prefetch_next=True)
    m0 = S.stats["prefetch_misses"]
    second = S.ranking_sampler_sbpr(d, dict(SPu), nr, B, prefetch_next=False)
    assert S.stats["prefetch_misses"] == m0 + 1
    np.random.seed(seed)
    S.drop_prefetch()
    a = S.ranking_sampler_sbpr(d, SPu, nr, B)
    b = S.ranking_sampler_sbpr(d, SPu, nr, B)
    assert same([first, second], [a, b])
    S.drop_prefetch()


def test_csr_builders_are_vectorised_and_equal_the_row_by_row_definition():
    """RankingData's CSR forms (samplers' iteration order, seen filter by user id with sorted rows, test rows in list order) and
    csr_from_dict against the row-by-row definition on random dicts (keys ascending and not, empty rows, absent users)."""
    g = np.random.Generator(np.random.PCG64(5))
    for trial in range(25):
        U, I = int(g.integers(1, 60)), int(g.integers(2, 50))
        ui = {int(u): [int(x) for x in g.choice(I, size=int(g.integers(0, min(I, 9))), replace=False)]
              for u in g.permutation(U)[:int(g.integers(1, U + 1))]}
        if trial % 2:
            ui = dict(sorted(ui.items()))
        te = {u: v[::-1][:2] for u, v in ui.items() if v}
        k, p, it = csr_from_dict(ui)
        assert list(k) == list(ui.keys()) and all(it[p[n]:p[n + 1]].tolist() == v for n, v in enumerate(ui.values()))
        for srt in (False, True):
            k, p, it = csr_from_dict(ui, n_rows=U, sort_rows=srt)
            for u in range(U):
                want = ui.get(u, [])
                assert it[p[u]:p[u + 1]].tolist() == (sorted(want) if srt else want)
        d = RankingData(U, I, ui, te)
        for u in range(U):
            assert d.seen_csr[1][d.seen_csr[0][u]:d.seen_csr[0][u + 1]].tolist() == sorted(ui.get(u, []))
            assert d.test_csr[1][d.test_csr[0][u]:d.test_csr[0][u + 1]].tolist() == te.get(u, [])


def test_lazy_descriptor_fields_round_trip(tmp_path):
    """csrc/lazy_desc.h: the packed int4 the lazy step reads per (batch position, role) keeps every field at its largest value
    (batches of 2^19 triples: k, start < 2^20, cnt <= 2^20, ci <= 2^19, hmod < 64) and at random ones -- host build, no GPU."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("needs hipcc")
    src = tmp_path / "t.hip"
    src.write_text(r'''
#include "lazy_desc.h"
#include <cstdio>
#include <cstdlib>
static int check(uint32_t row, bool single, int k, int start, int cnt, int ci, int hmod)
{
    const int4 d = lazy_desc_pack(row, single, k, start, cnt, ci, hmod);
    const bool ok = (uint32_t)(d.x & 0x7fffffff) == row && (d.x < 0) == single && lazy_desc_k(d) == k && lazy_desc_start(d) == start &&
                    lazy_desc_cnt(d) == cnt && lazy_desc_ci(d) == ci && lazy_desc_hmod(d) == hmod;
    if (!ok)
        printf("FAIL row %u single %d k %d start %d cnt %d ci %d hmod %d\n", row, (int)single, k, start, cnt, ci, hmod);
    return ok ? 0 : 1;
}
int main()
{
    int bad = 0;
    bad += check(0x7fffffffu, true, (1 << 20) - 1, (1 << 20) - 1, 1 << 20, 1 << 19, 63);
    bad += check(0u, false, 0, 0, 1, 0, 0);
    bad += check(12345u, false, 65536, 65535, 65537, 40000, 15);
    srand(7);
    for (int t = 0; t < 200000; ++t)
        bad += check((uint32_t)rand() & 0x7fffffffu, rand() & 1, rand() % (1 << 20), rand() % (1 << 20), 1 + rand() % (1 << 20), rand() % ((1 << 19) + 1), rand() % 64);
    printf("bad %d\n", bad);
    return bad != 0;
}
''')
    exe = str(tmp_path / "t")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-x", "hip", "-O1", "-I" + os.path.join(ROOT, "cleverrec_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
                    str(src), "-o", exe], check=True, timeout=600)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "bad 0" in out.stdout, out.stdout[-500:]


def test_lazy_index_restatement_against_plain_counting():
    """oracle/lazy_index.py (the checker of the device index builders) against plain counting: every slot's row count, its segment,
    its rank among the row's slots in batch order (positives before negatives), the processing order a permutation with the front
    rows' positive slots first, row after row."""
    from collections import Counter
    from oracle import lazy_index
    g = np.random.Generator(np.random.PCG64(17))
    for trial in range(12):
        nb, U, I, spw = int(g.integers(1, 400)), int(g.integers(2, 60)), int(g.integers(2, 40)), int(g.choice([16, 64]))
        u, i, j = g.integers(0, U, nb), np.minimum(g.zipf(1.4, nb) - 1, I - 1), g.integers(0, I, nb)
        ud, idd, jd, pu, (n, nf) = lazy_index.batch_index(u, i, j, spw)
        assert n == nb and sorted(pu[:, 0].tolist()) == list(range(nb)) and (pu[:, 1] == u[pu[:, 0]]).all()
        cu, cq, cpos = Counter(u.tolist()), Counter(i.tolist() + j.tolist()), Counter(i.tolist())
        unpack = lambda d: (int(d[0]) & 0x7fffffff, int(d[0]) < 0, int(d[1]) & 0xfffff, (int(d[1]) >> 20) & 0xfff, int(d[2]) & 0xfffff,  # noqa: E731
                            int(d[3]) & 0x1fffff, ((int(d[2]) >> 20) & 0xfff) | ((int(d[3]) >> 21) << 12))
        for pos in range(nb):
            row, single, k, _, start, cnt, _ = unpack(ud[pos])
            assert row == u[pos] and cnt == cu[row] and single == (cnt == 1) and k - start == int((u[:pos] == row).sum())
        front_seen = []
        for sl in range(nb):
            pos = int(pu[sl, 0])
            row, single, k, hmod, start, cnt, ci = unpack(idd[sl])
            assert row == i[pos] and cnt == cq[row] and k - start == int((i[:pos] == row).sum())
            assert ci == (cpos[row] if cpos[row] >= 2 else 0)
            if sl < nf:
                assert ci >= 2 and hmod == (sl - (k - start)) % spw
                front_seen.append(row)
            else:
                assert ci == 0 and hmod == 0
            row, single, k, _, start, cnt, cij = unpack(jd[sl])
            assert row == j[pos] and cnt == cq[row] and k - start == cpos[row] + int((j[:pos] == row).sum())
        assert front_seen == sorted(front_seen) and nf == sum(c for c in cpos.values() if c >= 2)
