"""utils/sampler.py of the reference, same function names and return tuples, backed by the
bit-exact C samplers of libcrec.so.  The reference draws from NumPy's global RandomState; to stay
a drop-in, each call takes np.random's MT19937 state, samples, and puts the advanced state back,
so any other np.random user sees exactly the stream it would have seen after the reference's
sampler."""
import math

import numpy as np

from .. import _lib

_ctx = None


def _host_ctx():
    global _ctx
    if _ctx is None:
        _ctx = _lib.Context(None)
    return _ctx


def _csr(data):
    return data.train_csr if hasattr(data, "train_csr") else _dict_csr(data.ui_train)


def _dict_csr(ui):
    from ..data import csr_from_dict
    return csr_from_dict(ui)


def _with_numpy_stream(fn):
    ctx = _host_ctx()
    st = np.random.get_state()
    assert st[0] == "MT19937"
    ctx.set_state(st[1], st[2])
    out = fn(ctx)
    key, pos = ctx.get_state()
    # has_gauss/cached_gaussian untouched: the samplers draw no normals
    np.random.set_state(("MT19937", key, pos, st[3], st[4]))
    return out


def pairwise_ranking_sampler(data, neg_ratio, batch_size, fism_like=False):
    """utils/sampler.py:46-74 -> (train_batches, u, i, j); int32 columns (the dtype the graph is fed)."""
    if fism_like:
        raise NotImplementedError("fism_like models are outside this core (SURVEY.md 2.1 #10)")
    u, i, j = _with_numpy_stream(lambda c: c.sample_pairwise(_csr(data), data.item_nums, neg_ratio))
    return math.ceil(len(u) / batch_size), u, i, j


def pointwise_ranking_sampler(data, neg_ratio, batch_size, fism_like=False):
    """utils/sampler.py:10-43 -> (train_batches, u, i, labels)."""
    if fism_like:
        raise NotImplementedError("fism_like models are outside this core (SURVEY.md 2.1 #10)")
    u, i, y = _with_numpy_stream(lambda c: c.sample_pointwise(_csr(data), data.item_nums, neg_ratio))
    return math.ceil(len(u) / batch_size), u, i, y


def ranking_sampler_cml(data, neg_ratio, batch_size):
    """utils/sampler.py:77-99 -> (train_batches, u, i, neg_items[N, neg_ratio])."""
    u, i, neg = _with_numpy_stream(lambda c: c.sample_cml(_csr(data), data.item_nums, neg_ratio))
    return math.ceil(len(u) / batch_size), u, i, neg
