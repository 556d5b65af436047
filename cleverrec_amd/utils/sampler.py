"""utils/sampler.py of the reference, same function names and return tuples, backed by the
bit-exact C samplers of libcrec.so.  The reference draws from NumPy's global RandomState; to stay
a drop-in, each call takes np.random's MT19937 state, samples, and puts the advanced state back,
so any other np.random user sees exactly the stream it would have seen after the reference's
sampler.

Prefetch (on by default, `sampler_prefetch=False` in the .properties file turns it off): the reference samples at the
top of every epoch, serially, with the device idle (RankingRecommender.py:37).  Here the NEXT epoch's sample is
drawn on a host thread while the GPU runs the current epoch -- from a private copy of the MT19937 state as the
current epoch's sampler left it.  At the next call the prefetched sample is used only if np.random still stands
exactly there (nobody drew in between; nothing on the path does, RankingRecommender.py:402-418); np.random then
jumps to the state after that sample.  Otherwise the prefetch is dropped and the call samples on the spot.  Either
way the returned arrays and the np.random state afterwards are bit-identical to the serial order."""
import math
import threading

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


def ranking_sampler_sbpr(data, SPu, neg_ratio, batch_size, is_suk=True, prefetch_next=False):
    """utils/sampler.py:102-141 -> (train_batches, u, i, i_s, i_neg[, suk]).  SPu: utils/tools.get_SPu(data) (list order matters).
    prefetch_next (extension): start drawing the next call's sample on a host thread (module docstring)."""
    from .tools import spu_csr
    cache = getattr(data, "_spu_csr", None)
    if cache is None or cache[0] is not SPu:
        cache = (SPu, spu_csr(data, SPu))
        try:
            data._spu_csr = cache
        except AttributeError:
            pass
    u, i, k, j, suk = _sample("sbpr", data, neg_ratio, prefetch_next, extra=(cache[1],), tag=id(SPu))
    nb = math.ceil(len(u) / batch_size)
    return (nb, u, i, k, j, suk) if is_suk else (nb, u, i, k, j)


_KINDS = {"pairwise": "sample_pairwise", "pointwise": "sample_pointwise", "cml": "sample_cml", "sbpr": "sample_sbpr"}


class _Prefetch:
    """One sample drawn ahead on a host thread from a private copy of the MT19937 state (see the module docstring)."""

    def __init__(self, kind, data, neg_ratio, key, pos, extra=(), tag=None):
        self.sig = (kind, id(data), int(neg_ratio), tag)
        self.key, self.pos = np.array(key, dtype=np.uint32, copy=True), int(pos)
        self.out = self.end = self.err = None
        csr, item_nums = _csr(data), data.item_nums

        def work():
            try:
                ctx = _lib.Context(None)  # its own host ctx: the RNG state is per ctx
                ctx.set_state(self.key, self.pos)
                self.out = getattr(ctx, _KINDS[kind])(csr, item_nums, neg_ratio, *extra)  # ctypes drops the GIL for the call
                self.end = ctx.get_state()
                ctx.close()
            except Exception as e:  # noqa: BLE001 -- surfaces as a fall-back to the serial path
                self.err = e
        self.thread = threading.Thread(target=work, daemon=True)
        self.thread.start()

    def matches(self, sig, st):
        return self.sig == sig and st[2] == self.pos and np.array_equal(st[1], self.key)


_pending = None
stats = {"prefetch_hits": 0, "prefetch_misses": 0}


def _sample(kind, data, neg_ratio, prefetch_next, extra=(), tag=None):
    """extra: further arguments of the ctx method (SBPR: the SPu CSR); tag: what else the sample depends on (id of the SPu dict)."""
    global _pending
    sig = (kind, id(data), int(neg_ratio), tag)
    st = np.random.get_state()
    assert st[0] == "MT19937"
    pf, _pending = _pending, None
    out = None
    if pf is not None and pf.matches(sig, st):
        pf.thread.join()
        if pf.err is None:
            out = pf.out
            np.random.set_state(("MT19937", pf.end[0], pf.end[1], st[3], st[4]))
            stats["prefetch_hits"] += 1
    if out is None:
        if pf is not None:
            stats["prefetch_misses"] += 1
        out = _with_numpy_stream(lambda c: getattr(c, _KINDS[kind])(_csr(data), data.item_nums, neg_ratio, *extra))
    if prefetch_next:
        st = np.random.get_state()
        _pending = _Prefetch(kind, data, neg_ratio, st[1], st[2], extra, tag)
    return out


def drop_prefetch(join=False):
    """forget a sample drawn ahead; np.random is untouched by construction.  join: wait for its host thread (end of
    training: nothing of the sampler keeps running, or holds the data, behind the caller's back)"""
    global _pending
    pf, _pending = _pending, None
    if join and pf is not None:
        pf.thread.join()


def pairwise_ranking_sampler(data, neg_ratio, batch_size, fism_like=False, prefetch_next=False):
    """utils/sampler.py:46-74 -> (train_batches, u, i, j); int32 columns (the dtype the graph is fed).
    prefetch_next (extension): start drawing the next call's sample on a host thread."""
    if fism_like:
        raise NotImplementedError("fism_like models are outside this core (SURVEY.md 2.1 #10)")
    u, i, j = _sample("pairwise", data, neg_ratio, prefetch_next)
    return math.ceil(len(u) / batch_size), u, i, j


def pointwise_ranking_sampler(data, neg_ratio, batch_size, fism_like=False, prefetch_next=False):
    """utils/sampler.py:10-43 -> (train_batches, u, i, labels)."""
    if fism_like:
        raise NotImplementedError("fism_like models are outside this core (SURVEY.md 2.1 #10)")
    u, i, y = _sample("pointwise", data, neg_ratio, prefetch_next)
    return math.ceil(len(u) / batch_size), u, i, y


def ranking_sampler_cml(data, neg_ratio, batch_size, prefetch_next=False):
    """utils/sampler.py:77-99 -> (train_batches, u, i, neg_items[N, neg_ratio])."""
    u, i, neg = _sample("cml", data, neg_ratio, prefetch_next)
    return math.ceil(len(u) / batch_size), u, i, neg
