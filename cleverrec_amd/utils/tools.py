"""Host helpers with the reference's names (utils/tools.py:9-87): re_index, timer, get_logger,
get_initializer.  get_loss / get_optimizer have no host object here: loss_func and optimizer
are parsed by the model class and carried out inside the HIP kernels."""
import functools
import logging
import math
import os
import sys
import time

import torch


def re_index(data_set):
    """utils/tools.py:9-15: ids in iteration order -> 0..n-1."""
    return {d: n for n, d in enumerate(data_set)}


def timer(text):
    """utils/tools.py:18-28 (same console lines)."""
    def decorator(func):
        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            t1 = time.time()
            print("Start %s..." % text)
            res = func(*args, **kwargs)
            print("%s done, time: %s" % (text, time.strftime("%H:%M:%S", time.gmtime(time.time() - t1))))
            return res
        return wrapper
    return decorator


def get_logger(log_dir, model):
    """utils/tools.py:31-48: root logger -> <log_dir>/<model>.log + stdout, same format."""
    os.makedirs(log_dir, exist_ok=True)
    logger = logging.getLogger()
    logger.setLevel(logging.DEBUG)
    fmt = logging.Formatter("%(asctime)s  %(message)s", datefmt="%Y-%m-%d %H:%M:%S")
    for h in (logging.FileHandler(os.path.join(log_dir, model + ".log")), logging.StreamHandler(sys.stdout)):
        h.setLevel(logging.DEBUG)
        h.setFormatter(fmt)
        logger.addHandler(h)
    return logger


def get_initializer(init_method, stddev=None, generator=None):
    """utils/tools.py:51-63.  Returns f(shape) -> fp32 CPU tensor.  'xavier_uniform' (used by the
    shipped GMF/MLP/NeuMF confs but unhandled by the reference) is Glorot-uniform.
    TF's own Philox stream is not reproducible (and the reference never seeds it); parity tests
    inject tables instead."""
    init_method = init_method.strip()

    def fans(shape):
        if len(shape) < 1:
            return 1, 1
        if len(shape) == 1:
            return shape[0], shape[0]
        return shape[-2], shape[-1]

    def f(shape):
        shape = tuple(int(s) for s in shape)
        g = generator
        if init_method == "normal":
            return torch.empty(shape).normal_(0.0, stddev, generator=g)
        if init_method == "tnormal":  # truncated at 2 sigma, redrawn
            return torch.nn.init.trunc_normal_(torch.empty(shape), 0.0, stddev, -2 * stddev, 2 * stddev, generator=g)
        if init_method == "uniform":
            return torch.empty(shape).uniform_(-stddev, stddev, generator=g)
        fi, fo = fans(shape)
        if init_method in ("xavier", "xavier_uniform"):
            lim = math.sqrt(6.0 / (fi + fo))
            return torch.empty(shape).uniform_(-lim, lim, generator=g)
        if init_method == "xavier_normal":
            # tf.contrib.layers.xavier_initializer(uniform=False) = variance_scaling_initializer(factor=1, FAN_AVG):
            # truncated_normal(0, sqrt(1.3 * factor / n)), n = (fan_in + fan_out) / 2 -- the 1.3 makes up for the truncation
            sd = math.sqrt(1.3 * 2.0 / (fi + fo))
            return torch.nn.init.trunc_normal_(torch.empty(shape), 0.0, sd, -2 * sd, 2 * sd, generator=g)
        raise ValueError("init_method=%r" % init_method)
    return f


def get_SPu(data, friends=None):
    """utils/tools.py:116-127 (friends = data.user_friends) and the SPu half of get_topK_friends_and_SPu (:196-206, friends =
    the top-K latent friends): per user the items its friends consumed and it did not, as a LIST -- in the order CPython
    iterates the set the reference builds, because ranking_sampler_sbpr indexes that list with the drawn position
    (utils/sampler.py:114-116).  The set arithmetic is therefore the reference's, operation for operation (the
    difference is taken after every friend, which changes the order elements enter the set)."""
    friends = data.user_friends if friends is None else friends
    SPu = {}
    for u in data.ui_train:
        u_s_items = set()
        if u in friends:
            for friend in friends[u]:
                if friend not in data.ui_train:
                    continue
                u_s_items = u_s_items.union(set(data.ui_train[friend])).difference(set(data.ui_train[u]))
            if u_s_items:
                SPu[u] = list(u_s_items)
    return SPu


def cunet(data):
    """The collaborative user network of utils/tools.py:132-141: users are linked when they share training items, the link
    weighs the number of shared items.  -> (nbr_weight: user -> {neighbour: weight}, draw_list: user -> int64 array holding
    every neighbour `weight` times), both in ui_train's iteration order -- the order the walks' tie-breaks and draws see.
    One sparse product instead of the reference's U^2 set intersections."""
    import numpy as np
    import scipy.sparse as sp
    users = list(data.ui_train)
    pos = {u: n for n, u in enumerate(users)}
    rows = np.concatenate([np.full(len(set(data.ui_train[u])), pos[u], np.int64) for u in users]) if users else np.zeros(0, np.int64)
    cols = np.concatenate([np.fromiter(set(data.ui_train[u]), np.int64) for u in users]) if users else np.zeros(0, np.int64)
    A = sp.csr_matrix((np.ones(len(rows), np.int64), (rows, cols)), shape=(len(users), data.item_nums))
    W = (A @ A.T).tocsr()
    W.sort_indices()  # ascending position = ui_train order
    nbr_weight, draw_list = {}, {}
    for n, u in enumerate(users):
        idx, w = W.indices[W.indptr[n]:W.indptr[n + 1]], W.data[W.indptr[n]:W.indptr[n + 1]]
        keep = (idx != n) & (w > 0)
        if keep.any():
            ids = [users[x] for x in idx[keep]]
            nbr_weight[u] = dict(zip(ids, (int(x) for x in w[keep])))
            draw_list[u] = np.repeat(np.asarray(ids, np.int64), w[keep])
    return nbr_weight, draw_list


def deep_walks(nbr_weight, draw_list, walk_count, walk_length):
    """utils/tools.py:145-167: walk_count walks of walk_length nodes from every user of the network, on the np.random stream.
    A step goes to the heaviest neighbour (first of equals, in the order CPython iterates the difference of the two sets)
    outside the `visited` record of the node it stands on; when there is none it draws one from the weighted list
    (np.random.choice).  The record is written under the walk's START user and read under the CURRENT node, as in the
    reference (:154,163).  The walks are shuffled as a Python list (np.random.shuffle's generic path)."""
    import numpy as np
    from collections import defaultdict
    walks, visited = [], defaultdict(dict)
    for start in draw_list:
        for _ in range(walk_count):
            path, here = [str(start)], start
            for _ in range(1, walk_length):
                fresh = list(set(nbr_weight[here].keys()).difference(set(visited[here])))
                if fresh:
                    weights = [nbr_weight[here][x] for x in fresh]
                    nxt = fresh[weights.index(max(weights))]
                else:
                    nxt = np.random.choice(draw_list[here])
                path.append(str(nxt))
                visited[start][nxt] = 1
                here = nxt
            walks.append(path)
    np.random.shuffle(walks)
    return walks


def walk_embeddings(walks, size, window, epochs=3, seed=1):
    """word2vec over the walks (utils/tools.py:173: gensim's Word2Vec(walks, size=, window=, min_count=0, iter=3), i.e. its
    defaults: CBOW with the context mean, 5 negative samples from the unigram^0.75 table, alpha 0.025 -> 0.0001, frequent-word
    subsampling at 1e-3, reduced windows).  gensim is used when it is importable; it is not in this image, so the same
    algorithm runs here in NumPy (positions in blocks of 512, updates accumulated per block).  Either way the vectors are
    not reproducible across implementations -- gensim's own differ from run to run with its default 3 worker threads.
    -> {token: float32 vector}."""
    import numpy as np
    try:
        from gensim.models import word2vec  # pragma: no cover -- absent here
        try:
            model = word2vec.Word2Vec(walks, size=size, window=window, min_count=0, iter=epochs)
        except TypeError:  # gensim >= 4 renamed the arguments
            model = word2vec.Word2Vec(walks, vector_size=size, window=window, min_count=0, epochs=epochs)
        return {t: np.asarray(model.wv[t], np.float32) for w in walks for t in w}
    except ImportError:
        pass
    g = np.random.Generator(np.random.PCG64(seed))  # private: the np.random stream the samplers continue on is not touched
    vocab = {}
    for w in walks:
        for t in w:
            vocab[t] = vocab.get(t, 0) + 1
    toks = list(vocab)
    tid = {t: n for n, t in enumerate(toks)}
    cnt = np.asarray([vocab[t] for t in toks], np.float64)
    V, total = len(toks), cnt.sum()
    syn0 = ((g.random((V, size)) - 0.5) / size).astype(np.float32)
    syn1 = np.zeros((V, size), np.float32)
    cdf = np.cumsum(cnt ** 0.75)
    cdf /= cdf[-1]
    thr = 1e-3 * total
    keep_p = np.minimum(1.0, (np.sqrt(cnt / thr) + 1) * (thr / cnt))
    sent = [np.asarray([tid[t] for t in w], np.int64) for w in walks]
    n_words, done, neg, BLK = sum(len(x) for x in sent) * epochs, 0, 5, 512
    for _ in range(epochs):
        # every (centre, context set) of the epoch, sentence by sentence, then processed in blocks
        centres, ctx_rows, ctx_cols = [], [], []
        n = 0
        for x in sent:
            x = x[g.random(len(x)) < keep_p[x]]
            L = len(x)
            if L < 2:
                continue
            red = g.integers(0, window, L)  # reduced window b: the span is window - b on either side
            for c in range(L):
                lo, hi = max(0, c - (window - red[c])), min(L, c + (window - red[c]) + 1)
                ctx = [x[k] for k in range(lo, hi) if k != c]
                if not ctx:
                    continue
                centres.append(x[c])
                ctx_rows.extend([n] * len(ctx))
                ctx_cols.extend(ctx)
                n += 1
        centres, ctx_rows, ctx_cols = np.asarray(centres, np.int64), np.asarray(ctx_rows, np.int64), np.asarray(ctx_cols, np.int64)
        starts = np.searchsorted(ctx_rows, np.arange(0, n + BLK, BLK))
        for b in range(0, n, BLK):
            alpha = max(0.0001, 0.025 - (0.025 - 0.0001) * done / max(1, n_words))
            e = min(n, b + BLK)
            r0, r1 = starts[b // BLK], starts[b // BLK + 1]
            rows, cols = ctx_rows[r0:r1] - b, ctx_cols[r0:r1]
            k = np.bincount(rows, minlength=e - b).astype(np.float32)
            h = np.zeros((e - b, size), np.float32)
            np.add.at(h, rows, syn0[cols])
            h /= k[:, None]
            tgt = np.concatenate([centres[b:e, None], np.searchsorted(cdf, g.random((e - b, neg)))[:, :]], 1).clip(max=V - 1)
            lab = np.zeros((e - b, neg + 1), np.float32)
            lab[:, 0] = 1.0
            lab[:, 1:][tgt[:, 1:] == tgt[:, :1]] = np.nan  # a negative equal to the centre is skipped
            f = 1.0 / (1.0 + np.exp(-np.einsum("bd,bnd->bn", h, syn1[tgt])))
            gr = np.nan_to_num((lab - f) * alpha).astype(np.float32)
            dh = np.einsum("bn,bnd->bd", gr, syn1[tgt])
            np.add.at(syn1, tgt.ravel(), (gr[:, :, None] * h[:, None, :]).reshape(-1, size))
            np.add.at(syn0, cols, (dh / k[:, None])[rows])  # cbow_mean: the error is shared out over the context words
            done += e - b
    return {t: syn0[tid[t]] for t in toks}


def get_topK_friends_and_SPu(data, walk_count, walk_length, walk_dim, window_size, topK_f):
    """utils/tools.py:130-207 (CUNE_BPR.py:20): collaborative user network -> deep walks -> word2vec user vectors -> per user
    the topK_f users of highest cosine similarity (stable order among equals, self excluded) -> SPu from those friends.
    -> (topK_friends: user -> list of user ids, SPu)."""
    import numpy as np
    nbr_weight, draw_list = cunet(data)
    vec = walk_embeddings(deep_walks(nbr_weight, draw_list, walk_count, walk_length), walk_dim, window_size)
    users = list(draw_list)
    E = np.stack([vec[str(u)] for u in users]).astype(np.float32) if users else np.zeros((0, walk_dim), np.float32)
    nrm = np.linalg.norm(E, axis=1)
    topK_friends = {}
    for n, u in enumerate(users):
        sims = E @ E[n] / (nrm * nrm[n])
        order = np.argsort(-np.delete(sims, n), kind="stable")[:topK_f]  # positions in the list without u itself
        topK_friends[u] = [users[x + (x >= n)] for x in order]
    return topK_friends, get_SPu(data, topK_friends)


def spu_csr(data, SPu):
    """(spu_ptr, spu_items, spu_suk) for crec_sample_sbpr: the SPu lists as rows parallel to ui_train's iteration order,
    and per entry the social coefficient suk = the number of the user's friends (with training items) who consumed it
    (utils/sampler.py:122-129)."""
    import numpy as np
    ptr, items, suk = [0], [], []
    sets = {}
    for u in data.ui_train:
        lst = SPu.get(u, [])
        friends = [f for f in data.user_friends.get(u, []) if f in data.ui_train]
        for f in friends:
            if f not in sets:
                sets[f] = set(data.ui_train[f])
        for it in lst:
            items.append(it)
            suk.append(sum(1 for f in friends if it in sets[f]))  # a friend listed twice counts twice, as in the reference's loop
        ptr.append(len(items))
    return (np.asarray(ptr, np.int64), np.asarray(items if items else [0], np.int32)[:len(items)],
            np.asarray(suk if suk else [0], np.int32)[:len(suk)])


def get_sp_mat(data):
    """utils/tools.py:100-113 (get_sp_mat, TransCF): the row-normalised user-item matrix and its row-normalised transpose as
    two CSRs instead of tf.SparseTensors: (ui_ptr [U+1] int64, ui_items int32, iu_ptr [I+1] int64, iu_users int32).
    Row r of the first holds ui_train[r] in list order (every entry weighs 1/len(ui_train[r])); row i of the second the users
    that hold item i, in ui_train's iteration order (every entry weighs 1/count).  The weights themselves are implied by
    the row lengths (float32(1.0 / n), the SparseTensor's values) and formed on the device."""
    import numpy as np
    U, I = data.user_nums, data.item_nums
    ui_len = np.zeros(U, dtype=np.int64)
    for u, items in data.ui_train.items():
        ui_len[u] = len(items)
    ui_ptr = np.zeros(U + 1, dtype=np.int64)
    np.cumsum(ui_len, out=ui_ptr[1:])
    ui_items = np.empty(int(ui_ptr[-1]), dtype=np.int32)
    us, its = [], []
    for u, items in data.ui_train.items():
        ui_items[ui_ptr[u]:ui_ptr[u + 1]] = np.asarray(items, dtype=np.int32)
        us.append(np.full(len(items), u, dtype=np.int32))
        its.append(np.asarray(items, dtype=np.int32))
    us = np.concatenate(us) if us else np.zeros(0, np.int32)
    its = np.concatenate(its) if its else np.zeros(0, np.int32)
    order = np.argsort(its, kind="stable")  # by item, ties in ui_train's iteration order
    iu_users = np.ascontiguousarray(us[order])
    iu_ptr = np.zeros(I + 1, dtype=np.int64)
    np.cumsum(np.bincount(its, minlength=I), out=iu_ptr[1:])
    return ui_ptr, ui_items, iu_ptr, iu_users
