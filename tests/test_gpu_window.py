"""Lazy-family steps with W consecutive batches per launch (option lazy_window: bpr_triples_lazy<.., WIN>): a triple
reads a row only after the last earlier batch of the launch that touches it has finished it, so the result must equal the launch
per batch BIT FOR BIT -- tables, optimizer slots and losses -- for every window length, optimizer and shape, ragged tails and
tables where every row is hot included.  Reference: model/RankingRecommender.py:39-47 (one sess.run per batch, in order)."""
import numpy as np
import pytest

from cleverrec_amd import runtime

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sess():
    s = runtime.Session(0)
    yield s
    s.close()


def _stream(seed, U, I, n, zipf=1.1):
    g = np.random.Generator(np.random.PCG64(seed))
    u = g.integers(0, U, n).astype(np.int32)
    i = np.minimum(g.zipf(zipf, n) - 1, I - 1).astype(np.int32)
    j = g.integers(0, I, n).astype(np.int32)
    return u, i, j


def _run(sess, W, mode, U, I, d, B, u, i, j, calls=1):
    g = np.random.Generator(np.random.PCG64(9))
    P0, Q0 = (g.standard_normal((U, d)) * 0.01).astype(np.float32), (g.standard_normal((I, d)) * 0.01).astype(np.float32)
    sess.ctx.set_option("lazy_window", W)
    try:
        st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode=mode)
        nb = -(-len(u) // B)
        cuts = [round(c * nb / calls) * B for c in range(calls)] + [len(u)]
        du, di, dj = (sess.to_dev(x) for x in (u, i, j))
        losses = [st.train_epoch(du[a:b], di[a:b], dj[a:b], B).cpu().numpy() for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        return [np.concatenate(losses)] + [getattr(st, n).cpu().numpy() for n in ("P", "Q", "mP", "mQ", "vP", "vQ")]
    finally:
        sess.ctx.set_option("lazy_window", 0)


CASES = [
    # U, I, d, B, batches, zipf, mode
    (200_000, 20_000, 128, 2048, 19, 1.1, "lazy"),     # S-big's shape in small: a Zipf head of hot items, cold users
    (200_000, 20_000, 64, 6144, 9, 1.1, "lazy"),
    (50_000, 3_000, 32, 1024, 33, 1.2, "lazy"),
    (300, 200, 128, 512, 21, 1.3, "lazy"),             # every row hot in every batch
    (300, 200, 20, 512, 11, 1.3, "lazy"),              # d no multiple of the lane group
    (40, 7, 16, 256, 13, 1.5, "lazy"),                 # rows with hundreds of slots (chunked rows, pieces) and a predecessor each
    (5_000, 1_000, 64, 700, 17, 1.2, "sgd"),
    (5_000, 1_000, 64, 700, 17, 1.2, "adagrad"),
    (100_000, 50_000, 256, 1000, 7, 1.1, "lazy"),
]


@pytest.mark.parametrize("U,I,d,B,nb,zipf,mode", CASES)
def test_window_equals_a_launch_per_batch_bit_for_bit(sess, U, I, d, B, nb, zipf, mode):
    u, i, j = _stream(U + d + B, U, I, nb * B + B // 3 + 1, zipf)
    ref = _run(sess, 0, mode, U, I, d, B, u, i, j)
    for W, calls in ((2, 1), (4, 2), (8, 1), (16, 3)):
        got = _run(sess, W, mode, U, I, d, B, u, i, j, calls)
        for x, y, name in zip(ref, got, ("loss", "P", "Q", "mP", "mQ", "vP", "vQ")):
            assert np.array_equal(x, y), "W = %d: %s" % (W, name)
