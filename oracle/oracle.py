"""ctypes front-end of oracle/crec_oracle.c plus small numpy restatements.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the
reference file:line it restates; paths are relative to the reference root.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

__all__ = [
    "build", "Rng", "to_csr", "dict_from_csr", "sample_pairwise", "sample_pointwise", "sample_cml", "reference_sampler",
    "AdamSchedule", "BprModel", "bpr_step", "scores", "topk", "rank_metrics",
    "cal_ranking_metrics_py", "test_model_rs_py", "train_model_py", "tf_softplus_np", "bpr_dp_local", "adam_dense", "test_model_loo_py", "bpr_shard_local", "adam_rows_lazy", "sharded_steps",
]


def build(force=False):
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "crec_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B", "liboracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_next_u32.restype = C.c_uint32
        _lib.orc_randint.restype = C.c_int64
        for f in ("orc_sample_pairwise", "orc_sample_pointwise", "orc_sample_cml"):
            getattr(_lib, f).restype = C.c_int64
    return _lib


def _p(a, t=None):
    return a.ctypes.data_as(C.c_void_p)


class _RngState(C.Structure):
    _fields_ = [("key", C.c_uint32 * 624), ("pos", C.c_int32)]


class Rng:
    """NumPy legacy global RNG (MT19937) as the samplers use it.

    Restates np.random.seed / randint / permutation, the only RNG calls on the
    path (utils/sampler.py:24,26,37,58,60,68,88,90,97)."""

    def __init__(self, seed=None):
        self.st = _RngState()
        if seed is not None:
            self.seed(seed)

    def seed(self, s):
        lib().orc_rng_seed(C.byref(self.st), C.c_uint32(s))

    def set_state(self, key, pos):
        key = np.ascontiguousarray(key, dtype=np.uint32)
        lib().orc_rng_set_state(C.byref(self.st), _p(key), C.c_int32(pos))

    def get_state(self):
        key = np.zeros(624, dtype=np.uint32)
        pos = C.c_int32()
        lib().orc_rng_get_state(C.byref(self.st), _p(key), C.byref(pos))
        return key, pos.value

    def next_u32(self):
        return int(lib().orc_next_u32(C.byref(self.st)))

    def randint(self, n):
        return int(lib().orc_randint(C.byref(self.st), C.c_int64(n)))

    def permutation(self, n):
        out = np.empty(n, dtype=np.int64)
        lib().orc_permutation(C.byref(self.st), C.c_int64(n), _p(out))
        return out


def to_csr(ui, n_rows=None):
    """dict[int -> list[int]] in iteration order -> (keys int32, ptr int64, items int32).

    With n_rows: a CSR indexed by row id (empty rows for absent keys)."""
    if n_rows is None:
        keys = np.fromiter(ui.keys(), dtype=np.int32, count=len(ui))
        lens = np.fromiter((len(v) for v in ui.values()), dtype=np.int64, count=len(ui))
        ptr = np.zeros(len(ui) + 1, dtype=np.int64)
        np.cumsum(lens, out=ptr[1:])
        items = np.fromiter((x for v in ui.values() for x in v), dtype=np.int32, count=int(ptr[-1]))
        return keys, ptr, items
    lens = np.zeros(n_rows, dtype=np.int64)
    for k, v in ui.items():
        lens[k] = len(v)
    ptr = np.zeros(n_rows + 1, dtype=np.int64)
    np.cumsum(lens, out=ptr[1:])
    items = np.zeros(int(ptr[-1]), dtype=np.int32)
    for k, v in ui.items():
        items[ptr[k]:ptr[k + 1]] = v
    return np.arange(n_rows, dtype=np.int32), ptr, items


def dict_from_csr(keys, ptr, items):
    return {int(k): [int(x) for x in items[ptr[n]:ptr[n + 1]]] for n, k in enumerate(keys)}


def _sample(fn, rng, ui_train, item_nums, neg_ratio, batch_size, per_pos, third_dtype, third_width):
    keys, ptr, items = to_csr(ui_train)
    n_max = int(ptr[-1]) * per_pos
    u = np.empty(max(n_max, 1), dtype=np.int64)
    i = np.empty(max(n_max, 1), dtype=np.int64)
    t = np.empty(max(n_max, 1) * third_width, dtype=third_dtype)
    n = getattr(lib(), fn)(C.byref(rng.st), _p(ptr), _p(items), C.c_int64(len(keys)), _p(keys),
                           C.c_int32(item_nums), C.c_int(neg_ratio), _p(u), _p(i), _p(t))
    assert n == n_max
    t = t[:n * third_width]
    if third_width > 1:
        t = t.reshape(n, third_width)
    return math.ceil(n / batch_size), u[:n], i[:n], t


def sample_pairwise(rng, ui_train, item_nums, neg_ratio, batch_size):
    """pairwise_ranking_sampler (utils/sampler.py:46-74) -> (train_batches, u, i, j) int64."""
    return _sample("orc_sample_pairwise", rng, ui_train, item_nums, neg_ratio, batch_size, neg_ratio, np.int64, 1)


def sample_pointwise(rng, ui_train, item_nums, neg_ratio, batch_size):
    """pointwise_ranking_sampler (utils/sampler.py:10-43) -> (train_batches, u, i, y float64)."""
    return _sample("orc_sample_pointwise", rng, ui_train, item_nums, neg_ratio, batch_size, 1 + neg_ratio,
                   np.float64, 1)


def sample_cml(rng, ui_train, item_nums, neg_ratio, batch_size):
    """ranking_sampler_cml (utils/sampler.py:77-99) -> (train_batches, u, i, neg[N, neg_ratio])."""
    return _sample("orc_sample_cml", rng, ui_train, item_nums, neg_ratio, batch_size, 1, np.int64, neg_ratio)


def reference_sampler(kind, ui_train, item_nums, neg_ratio):
    """The three samplers drawing from np.random's GLOBAL state, as the reference does (utils/sampler.py:10-99): reads the
    state, samples with the C restatement above, leaves np.random where the reference's call would.  -> (u, i, third)."""
    st = np.random.get_state()
    rng = Rng()
    rng.set_state(st[1], st[2])
    fn = {"pairwise": sample_pairwise, "pointwise": sample_pointwise, "cml": sample_cml}[kind]
    _, u, i, t = fn(rng, ui_train, item_nums, neg_ratio, 1 << 30)
    key, pos = rng.get_state()
    np.random.set_state(("MT19937", key, pos, st[3], st[4]))
    return u, i, t


class AdamSchedule:
    """tf.train.AdamOptimizer(lr) step-size schedule (utils/tools.py:83-84).

    lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t); beta powers are running fp32 products
    (TF1 keeps them as fp32 slot variables multiplied once per step)."""

    def __init__(self, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.beta1, self.beta2, self.eps = np.float32(lr), np.float32(beta1), np.float32(beta2), np.float32(eps)
        self.b1p, self.b2p = C.c_float(self.beta1), C.c_float(self.beta2)

    def next(self, n=1):
        out = np.empty(n, dtype=np.float32)
        lib().orc_adam_lr_table(C.c_float(self.lr), C.c_float(self.beta1), C.c_float(self.beta2), C.c_int64(0),
                                C.c_int64(n), C.byref(self.b1p), C.byref(self.b2p), _p(out))
        return out


class BprModel:
    """Parameters + Adam slots of model/ranking/BPR.py:23-29 (P, Q fp32) on the CPU."""

    def __init__(self, P, Q, lr=1e-3, reg=0.01, adam_mode=0):
        """adam_mode: 0 Adam (TF1: every row every step), 1 Adam on touched rows only, 2 SGD, 3 Adagrad (the accumulator
        lives in mP / mQ, initial_accumulator_value 0.1) -- utils/tools.py:79-87."""
        self.P = np.array(P, dtype=np.float32, order="C")
        self.Q = np.array(Q, dtype=np.float32, order="C")
        self.mP, self.vP = np.zeros_like(self.P), np.zeros_like(self.P)
        self.mQ, self.vQ = np.zeros_like(self.Q), np.zeros_like(self.Q)
        if adam_mode == 3:
            self.mP += np.float32(0.1)
            self.mQ += np.float32(0.1)
        self.sched = AdamSchedule(lr)
        self.reg, self.adam_mode = float(reg), int(adam_mode)
        self._gP, self._gQ = np.empty_like(self.P), np.empty_like(self.Q)
        self._cP, self._cQ = np.empty(len(self.P), np.int32), np.empty(len(self.Q), np.int32)

    def step(self, u, i, j, want_s=False):
        """One sess.run([train, loss]) (RankingRecommender.py:46): returns the summed batch loss."""
        lr_t = float(self.sched.next(1)[0])
        if self.adam_mode >= 2:
            lr_t = float(self.sched.lr)  # SGD / Adagrad: the plain learning rate
        return bpr_step(self, u, i, j, lr_t, want_s)


def bpr_step(m, u, i, j, lr_t, want_s=False):
    u = np.ascontiguousarray(u, dtype=np.int32)
    i = np.ascontiguousarray(i, dtype=np.int32)
    j = np.ascontiguousarray(j, dtype=np.int32)
    loss = C.c_double()
    s = np.empty(len(u), dtype=np.float32) if want_s else None
    U, d = m.P.shape
    lib().orc_bpr_step(_p(m.P), _p(m.Q), _p(m.mP), _p(m.vP), _p(m.mQ), _p(m.vQ), C.c_int64(U), C.c_int64(len(m.Q)),
                       C.c_int(d), _p(u), _p(i), _p(j), C.c_int64(len(u)), C.c_float(m.reg), C.c_float(lr_t),
                       C.c_float(m.sched.beta1), C.c_float(m.sched.beta2), C.c_float(m.sched.eps),
                       C.c_int(m.adam_mode), _p(m._gP), _p(m._gQ), _p(m._cP), _p(m._cQ), C.byref(loss),
                       _p(s) if want_s else None)
    return (loss.value, s) if want_s else loss.value


def train_model_py(m, batches, u, i, j, batch_size):
    """RankingRecommender.train_model (model/RankingRecommender.py:33-61): contiguous
    slices of the sampled epoch, mean of the per-batch summed losses."""
    total = 0.0
    per_batch = []
    for b in range(batches):
        sl = slice(b * batch_size, (b + 1) * batch_size)
        loss = m.step(u[sl], i[sl], j[sl])
        per_batch.append(loss)
        total += loss
    return total / batches, np.array(per_batch)


def scores(P, Q, users, order=0):
    """BPR._predict rs branch (model/ranking/BPR.py:51): P[users] @ Q.T in fp32."""
    P = np.ascontiguousarray(P, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    users = np.ascontiguousarray(users, dtype=np.int32)
    out = np.empty((len(users), len(Q)), dtype=np.float32)
    lib().orc_scores(_p(P), _p(Q), C.c_int64(len(Q)), C.c_int(P.shape[1]), _p(users), C.c_int64(len(users)),
                     C.c_int(order), _p(out))
    return out


def topk(sc, users, seen_ptr, seen_items, K, ascending=False):
    """argsort + seen filter of test_model_rs (model/RankingRecommender.py:221-240)."""
    sc = np.ascontiguousarray(sc, dtype=np.float32)
    users = np.ascontiguousarray(users, dtype=np.int32)
    ids = np.empty((len(users), K), dtype=np.int32)
    val = np.empty((len(users), K), dtype=np.float32)
    lib().orc_topk(_p(sc), C.c_int64(sc.shape[1]), _p(users), C.c_int64(len(users)),
                   _p(np.ascontiguousarray(seen_ptr, dtype=np.int64)),
                   _p(np.ascontiguousarray(seen_items, dtype=np.int32)), C.c_int(K), C.c_int(int(ascending)),
                   _p(ids), _p(val))
    return ids, val


def rank_metrics(topk_ids, users, ks, test_ptr, test_items):
    """cal_ranking_metrics over a user batch (utils/metrics.py:9-19) -> hr, mrr, ndcg [len(ks), Bt] float64."""
    topk_ids = np.ascontiguousarray(topk_ids, dtype=np.int32)
    users = np.ascontiguousarray(users, dtype=np.int32)
    ks = np.ascontiguousarray(ks, dtype=np.int32)
    Bt = len(users)
    hr, mrr, ndcg = (np.empty((len(ks), Bt), dtype=np.float64) for _ in range(3))
    lib().orc_rank_metrics(_p(topk_ids), _p(users), C.c_int64(Bt), C.c_int(topk_ids.shape[1]), _p(ks),
                           C.c_int(len(ks)), _p(np.ascontiguousarray(test_ptr, dtype=np.int64)),
                           _p(np.ascontiguousarray(test_items, dtype=np.int32)), _p(hr), _p(mrr), _p(ndcg))
    return hr, mrr, ndcg


# ---------------------------------------------------------------------------
# small pure-Python / numpy restatements used to cross-check the C versions
# ---------------------------------------------------------------------------
def cal_ranking_metrics_py(real_items, rec_items, K):
    """utils/metrics.py:9-19, statement for statement (float64)."""
    rec = list(rec_items)
    hit, mrr, dcg, idcg = 0, 0.0, 0.0, 0.0
    for idx_true, item in enumerate(real_items):
        if item in rec:
            hit += 1
            rank = rec.index(item)
            mrr += 1.0 / (rank + 1)
            dcg += 1.0 / math.log2(rank + 2)
        idcg += 1.0 / math.log2(idx_true + 2)
    return hit / min(K, len(real_items)), mrr, dcg / idcg


def test_model_rs_py(sc_fn, test_users, ui_train, ui_test, topk_list, batch_size_t, ascending=False):
    """test_model_rs (model/RankingRecommender.py:198-247) with a stable, lowest-id-first
    argsort.  sc_fn(users) -> [len(users), I] fp32 scores."""
    HR, MRR, NDCG = ({k: [] for k in range(len(topk_list))} for _ in range(3))
    Kmax = topk_list[-1]
    for t in range(math.ceil(len(test_users) / batch_size_t)):
        cur = test_users[t * batch_size_t:(t + 1) * batch_size_t]
        sc = sc_fn(cur)
        args = np.argsort(sc if ascending else -sc, axis=1, kind="stable")
        for r, uu in enumerate(cur):
            if uu not in ui_train:
                top = list(args[r][:Kmax])
            else:
                seen = set(ui_train[uu])
                top = [a for a in args[r] if a not in seen][:Kmax]
            for kid, K in enumerate(topk_list):
                h, m, n = cal_ranking_metrics_py(ui_test[uu], top[:K], K)
                HR[kid].append(h)
                MRR[kid].append(m)
                NDCG[kid].append(n)
    return HR, MRR, NDCG


def tf_softplus_np(z):
    """Eigen softplus as TF1 evaluates -log_sigmoid (utils/tools.py:70-71)."""
    z = np.asarray(z, dtype=np.float32)
    thr = np.float32(np.log(np.float32(1.1920929e-07)) + 2.0)
    return np.where(z > -thr, z, np.where(z < thr, np.exp(z), np.log1p(np.exp(z)))).astype(np.float32)


def bpr_dp_local(m, u, i, j, lr_t):
    """Data-parallel local half of a step (orc_bpr_dp_local): updates m.P in place, returns (dQ, loss)."""
    u, i, j = (np.ascontiguousarray(x, dtype=np.int32) for x in (u, i, j))
    dQ = np.empty_like(m.Q)
    loss = C.c_double()
    U, d = m.P.shape
    lib().orc_bpr_dp_local(_p(m.P), _p(m.Q), _p(m.mP), _p(m.vP), C.c_int64(U), C.c_int64(len(m.Q)), C.c_int(d), _p(u),
                           _p(i), _p(j), C.c_int64(len(u)), C.c_float(m.reg), C.c_float(lr_t), C.c_float(m.sched.beta1),
                           C.c_float(m.sched.beta2), C.c_float(m.sched.eps), _p(m._gP), _p(m._cP), _p(dQ), C.byref(loss))
    return dQ, loss.value


def adam_dense(m, g, lr_t):
    """tf1 Adam of every Q row from the summed gradient (orc_adam_dense)."""
    g = np.ascontiguousarray(g, dtype=np.float32)
    lib().orc_adam_dense(_p(m.Q), _p(m.mQ), _p(m.vQ), _p(g), C.c_int64(len(m.Q)), C.c_int(m.Q.shape[1]), C.c_float(lr_t),
                         C.c_float(m.sched.beta1), C.c_float(m.sched.beta2), C.c_float(m.sched.eps))


def bpr_shard_local(m, u, i, j, lr_t):
    """Row-sharded lazy step, one rank's half (orc_bpr_shard_local): lazy Adam on the rank's touched rows of m.P in
    place; returns (dQ [I,d] the rank's per-item gradient sums, cntQ [I] occurrences, loss)."""
    u, i, j = (np.ascontiguousarray(x, dtype=np.int32) for x in (u, i, j))
    dQ, cntQ = np.empty_like(m.Q), np.empty(len(m.Q), np.int32)
    loss = C.c_double()
    U, d = m.P.shape
    lib().orc_bpr_shard_local(_p(m.P), _p(m.Q), _p(m.mP), _p(m.vP), C.c_int64(U), C.c_int64(len(m.Q)), C.c_int(d), _p(u),
                              _p(i), _p(j), C.c_int64(len(u)), C.c_float(m.reg), C.c_float(lr_t), C.c_float(m.sched.beta1),
                              C.c_float(m.sched.beta2), C.c_float(m.sched.eps), _p(m._gP), _p(m._cP), _p(dQ), _p(cntQ),
                              C.byref(loss))
    return dQ, cntQ, loss.value


def adam_rows_lazy(m, g, touched, lr_t):
    """lazy Adam of the Q rows with touched != 0 from the rank-summed gradient (orc_adam_rows_lazy)."""
    g = np.ascontiguousarray(g, dtype=np.float32)
    touched = np.ascontiguousarray(touched, dtype=np.int32)
    lib().orc_adam_rows_lazy(_p(m.Q), _p(m.mQ), _p(m.vQ), _p(g), _p(touched), C.c_int64(len(m.Q)), C.c_int(m.Q.shape[1]),
                             C.c_float(lr_t), C.c_float(m.sched.beta1), C.c_float(m.sched.beta2), C.c_float(m.sched.eps))


def sharded_steps(P0, Q0, user_ranges, streams, B, lr=1e-3, reg=0.01):
    """The row-sharded epoch as one process would compute it (arithmetic order of include/crec.h "row-sharded
    training"): rank r owns users [lo, hi) = user_ranges[r] and runs its own triples streams[r] = (u local ids, i, j
    global item ids); per step every rank forms its per-item gradient sums against the step's Q, the sums are added
    over ranks in rank order, lazy Adam on the touched rows.  Returns (P full, Q, losses per step summed over ranks)."""
    G = len(user_ranges)
    ms = [BprModel(P0[lo:hi], Q0, lr=lr, reg=reg, adam_mode=1) for lo, hi in user_ranges]
    owner = BprModel(P0[:1], Q0, lr=lr, reg=reg, adam_mode=1)  # holds Q and its slots
    nb = max(-(-len(s[0]) // B) for s in streams)
    losses = []
    for b in range(nb):
        lr_t = float(owner.sched.next(1)[0])
        g, touched, tot = np.zeros_like(owner.Q), np.zeros(len(owner.Q), np.int32), 0.0
        for r in range(G):
            u, i, j = (x[b * B:(b + 1) * B] for x in streams[r])
            ms[r].Q = owner.Q  # every rank sees the step's Q
            dQ, cnt, loss = bpr_shard_local(ms[r], u, i, j, lr_t)
            g = g + dQ if r else dQ  # rank order; a rank that does not touch a row adds an exact 0
            touched += cnt
            tot += loss
        adam_rows_lazy(owner, g, touched, lr_t)
        losses.append(tot)
    P = np.concatenate([m.P for m in ms])
    return P, owner.Q, np.array(losses), owner


def test_model_loo_py(pair_score_fn, test_users, ui_test, neg_samples, topk_list, batch_size_t, ascending=False):
    """test_model_loo (model/RankingRecommender.py:250-299): pair_score_fn(u, items) -> fp32 scores of the
    listed (u, item) pairs; a stable argsort stands in for np.argsort."""
    HR, MRR, NDCG = ({k: [] for k in range(len(topk_list))} for _ in range(3))
    Kmax = topk_list[-1]
    for uu in test_users:
        items = ui_test[uu]
        sc = np.asarray(pair_score_fn(uu, items), dtype=np.float32)
        args = np.argsort(sc if ascending else -sc, kind="stable")[:Kmax]
        real = items[neg_samples:]
        for kid, K in enumerate(topk_list):
            rec = np.take(items, args[:K])
            h, m, n = cal_ranking_metrics_py(real, rec, K)
            HR[kid].append(h)
            MRR[kid].append(m)
            NDCG[kid].append(n)
    return HR, MRR, NDCG
