"""The `data` protocol of the hot path and shape-faithful synthetic datasets.

The reference's trainers consume an object with .user_nums, .item_nums, .ui_train, .ui_test
(dict[int -> list[int]], keys ascending; model/RankingPreprocess.py:17,43,117-118).  RankingData
carries the same fields plus CSR forms the kernels and the C sampler read directly.  MovieLens
files may not be redistributed and do not travel to the GPU box, so benchmarks and GPU tests run
on seeded synthetic data with MovieLens-like marginals."""
import numpy as np


def csr_from_dict(ui, n_rows=None, sort_rows=False):
    """dict -> (keys int32, ptr int64, items int32).  n_rows=None: rows in dict order (what the
    samplers iterate).  n_rows=U: one row per id, empty rows for absent keys.  No Python-level work per item: the lists
    are flattened by one chained fromiter, rows placed / sorted with array operations (a 10 M-user dict is minutes as loops)."""
    from itertools import chain
    k_in = np.fromiter(ui.keys(), dtype=np.int64, count=len(ui))
    l_in = np.fromiter((len(v) for v in ui.values()), dtype=np.int64, count=len(ui))
    flat = np.fromiter(chain.from_iterable(ui.values()), dtype=np.int32, count=int(l_in.sum()))
    if n_rows is None:
        ptr = np.zeros(len(k_in) + 1, dtype=np.int64)
        np.cumsum(l_in, out=ptr[1:])
        return k_in.astype(np.int32), ptr, flat
    keys = np.arange(n_rows, dtype=np.int32)
    lens = np.zeros(n_rows, dtype=np.int64)
    lens[k_in] = l_in
    ptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(lens, out=ptr[1:])
    row_of = np.repeat(k_in, l_in)  # the row of every flattened item
    if sort_rows:
        order = np.lexsort((flat, row_of))  # by row, items ascending inside a row
    else:
        order = np.argsort(row_of, kind="stable")  # by row, list order kept
    return keys, ptr, flat[order]


def dict_from_csr(keys, ptr, items):
    return {int(k): items[ptr[n]:ptr[n + 1]].tolist() for n, k in enumerate(keys)}


class RankingData:
    """user_nums / item_nums / ui_train / ui_test as RankingPreprocess publishes them, plus:
      train_csr       (user_ids, ptr, items) in ui_train iteration order   -> samplers
      seen_csr        (ptr[U+1], items sorted per user) by user id          -> eval seen filter
      test_csr        (ptr[U+1], items in list order) by user id            -> metrics
      test_users      list(ui_test.keys())                                  (RankingRecommender.py:23)
    Build from dicts (small data, tests) or straight from CSR arrays (large synthetic data: the
    dict views are then materialised lazily and only if someone asks)."""

    def __init__(self, user_nums, item_nums, ui_train=None, ui_test=None, train_csr=None, test_csr_rows=None):
        self.user_nums, self.item_nums = int(user_nums), int(item_nums)
        self._ui_train, self._ui_test = ui_train, ui_test
        if train_csr is None:
            train_csr = csr_from_dict(ui_train)
        self.train_csr = tuple(np.ascontiguousarray(a) for a in train_csr)
        if test_csr_rows is None:
            test_csr_rows = csr_from_dict(ui_test)
        self._test_rows = tuple(np.ascontiguousarray(a) for a in test_csr_rows)
        self.test_users = self._test_rows[0]
        self.seen_csr = self._by_user_id(self.train_csr, sort_rows=True)
        self.test_csr = self._by_user_id(self._test_rows, sort_rows=False)

    def _by_user_id(self, rows, sort_rows):
        """(keys, ptr, items) in iteration order -> (ptr[U+1], items) by user id, a row's items sorted or in list order; array
        operations only (the per-user loop this replaces took minutes at ten million users)."""
        keys, ptr, items = rows
        lens_in = np.diff(ptr)
        lens = np.zeros(self.user_nums, dtype=np.int64)
        lens[keys] = lens_in
        full = np.zeros(self.user_nums + 1, dtype=np.int64)
        np.cumsum(lens, out=full[1:])
        keys64 = np.asarray(keys, dtype=np.int64)
        ascending = len(keys64) < 2 or bool((np.diff(keys64) > 0).all())  # RankingPreprocess publishes its dicts with ascending keys
        items = np.asarray(items, dtype=np.int32)
        if not sort_rows:
            if ascending:
                return full, np.ascontiguousarray(items)
            return full, np.ascontiguousarray(items[np.argsort(np.repeat(keys64, lens_in), kind="stable")])
        comp = (np.repeat(keys64, lens_in) << 32) | items.astype(np.int64)  # (row, item) as one key: one sort, no gather (item ids are non-negative)
        comp.sort()
        return full, (comp & 0xffffffff).astype(np.int32)

    def loo_csr(self, neg_samples):
        """Views for test_model_loo (model/RankingRecommender.py:250-299), where ui_test[u] holds the sampled
        negatives first and the held-out items after them (model/RankingPreprocess.py:121-129):
          cand (ptr[U+1], items sorted per user): every listed item -- the only ones that compete;
          real (ptr[U+1], items): ui_test[u][neg_samples:], the ground truth."""
        keys, ptr, items = self._test_rows
        cand, real = {}, {}
        for n, k in enumerate(keys):
            seg = items[ptr[n]:ptr[n + 1]]
            cand[int(k)] = np.unique(seg)
            real[int(k)] = seg[neg_samples:]
        out = []
        for d in (cand, real):
            lens = np.zeros(self.user_nums, dtype=np.int64)
            for k, v in d.items():
                lens[k] = len(v)
            full = np.zeros(self.user_nums + 1, dtype=np.int64)
            np.cumsum(lens, out=full[1:])
            flat = np.zeros(max(int(full[-1]), 1), dtype=np.int32)
            for k, v in d.items():
                flat[full[k]:full[k + 1]] = v
            out.append((full, flat))
        return out[0], out[1]

    @property
    def ui_train(self):
        if self._ui_train is None:
            self._ui_train = dict_from_csr(*self.train_csr)
        return self._ui_train

    @property
    def ui_test(self):
        if self._ui_test is None:
            self._ui_test = dict_from_csr(*self._test_rows)
        return self._ui_test

    @property
    def n_train(self):
        return int(self.train_csr[1][-1])


# ---------------------------------------------------------------------------------------------
# synthetic datasets
# ---------------------------------------------------------------------------------------------
SHAPES = {
    # name: (users, items, interactions, min interactions per user)
    "S-ml100k": (943, 1349, 99286, 20),
    "S-ml1m": (6040, 3706, 1000208, 20),
}


def synth_interactions(U, I, n_inter, min_deg, seed=20240601, zipf_alpha=1.0, sigma=1.0):
    """(user, item) pairs without duplicates: user degree ~ log-normal, item popularity ~ Zipf."""
    g = np.random.Generator(np.random.PCG64(seed))
    raw = g.lognormal(0.0, sigma, U)
    cap = int(0.6 * I)
    deg = np.full(U, min_deg, dtype=np.int64)
    extra = n_inter - deg.sum()
    if extra < 0:
        raise ValueError("n_inter too small for min_deg")
    w = raw / raw.sum()
    for _ in range(20):  # distribute the remainder proportionally under the per-user cap
        room = cap - deg
        add = np.minimum(np.floor(w * extra).astype(np.int64), room)
        deg += add
        extra -= add.sum()
        if extra <= 0:
            break
        w = np.where(cap - deg > 0, raw, 0.0)
        if w.sum() == 0:
            break
        w = w / w.sum()
    k = 0
    while extra > 0:  # leftovers from the floors
        if deg[k % U] < cap:
            deg[k % U] += 1
            extra -= 1
        k += 1
    logw = -zipf_alpha * np.log(np.arange(1, I + 1))
    logw = logw[g.permutation(I)]  # popularity not aligned with item id
    users = np.repeat(np.arange(U, dtype=np.int32), deg)
    items = np.empty(deg.sum(), dtype=np.int32)
    pos = 0
    block = max(1, (1 << 24) // I)
    for u0 in range(0, U, block):  # Gumbel top-k = weighted sampling without replacement
        u1 = min(U, u0 + block)
        keys = logw[None, :] + g.gumbel(size=(u1 - u0, I))
        order = np.argsort(-keys, axis=1)
        for r in range(u1 - u0):
            d = deg[u0 + r]
            items[pos:pos + d] = order[r, :d]
            pos += d
    return users, items


def split_rs(users, items, U, I, ratios=(0.7, 0.2, 0.1), item_min=5, seed=0):
    """Random split in the spirit of RankingPreprocess (model/RankingPreprocess.py:35-39,108-118):
    drop items with < item_min ratings, re-index, shuffle, train = first r1, test = last r3;
    ui_train / ui_test keep the shuffled row order inside each user, users ascending."""
    if item_min > 0:
        cnt = np.bincount(items, minlength=I)
        keep = cnt[items] >= item_min
        users, items = users[keep], items[keep]
    uu, users = np.unique(users, return_inverse=True)
    ii, items = np.unique(items, return_inverse=True)
    U2, I2 = len(uu), len(ii)
    g = np.random.Generator(np.random.PCG64(seed))
    perm = g.permutation(len(users))
    n_train = int(round(ratios[0] * len(users)))
    n_test = int(round(ratios[2] * len(users)))
    tr, te = perm[:n_train], perm[len(users) - n_test:]

    def rows(sel):
        order = sel[np.argsort(users[sel], kind="stable")]
        us = users[order]
        keys, counts = np.unique(us, return_counts=True)
        ptr = np.zeros(len(keys) + 1, dtype=np.int64)
        np.cumsum(counts, out=ptr[1:])
        return keys.astype(np.int32), ptr, items[order].astype(np.int32)

    return RankingData(U2, I2, train_csr=rows(tr), test_csr_rows=rows(te))


def synthetic(name="S-ml1m", seed=20240601, split_seed=0):
    U, I, n, md = SHAPES[name]
    users, items = synth_interactions(U, I, n, md, seed)
    return split_rs(users, items, U, I, seed=split_seed)


def synthetic_big(U=10_000_000, I=1_000_000, per_user=50, seed=1):
    """S-big: per_user interactions per user, Zipf(1.0) items; CSR only.  Duplicate
    (u, i) pairs are not removed (they are rare at this density and do not change the byte traffic)."""
    g = np.random.Generator(np.random.PCG64(seed))
    w = 1.0 / np.arange(1, I + 1)
    cdf = np.cumsum(w / w.sum())
    n = U * per_user
    items = np.searchsorted(cdf, g.random(n)).astype(np.int32)
    np.minimum(items, I - 1, out=items)
    ptr = np.arange(U + 1, dtype=np.int64) * per_user
    return np.arange(U, dtype=np.int32), ptr, items
