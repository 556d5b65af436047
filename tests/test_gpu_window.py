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
        st.flush()  # (deferred Adam: every row brought up to the last step; a no-op otherwise)
        return [np.concatenate(losses)] + [getattr(st, n).cpu().numpy() for n in ("P", "Q", "mP", "mQ", "vP", "vQ")]
    finally:
        sess.ctx.set_option("lazy_window", -1)


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
    (200_000, 20_000, 64, 2048, 23, 1.1, "deferred"),  # TF1-exact Adam by replay: the rows' step stamps travel with them
    (300, 200, 32, 512, 40, 1.3, "deferred"),          # every row hot: short replays, every stamp rewritten in every batch
    (60_000, 900, 128, 4096, 9, 1.2, "deferred"),
]


@pytest.mark.parametrize("U,I,d,B,nb,zipf,mode", CASES)
def test_window_equals_a_launch_per_batch_bit_for_bit(sess, U, I, d, B, nb, zipf, mode):
    u, i, j = _stream(U + d + B, U, I, nb * B + B // 3 + 1, zipf)
    ref = _run(sess, 0, mode, U, I, d, B, u, i, j)
    for W, calls in ((2, 1), (4, 2), (8, 1), (12, 2), (16, 3)):
        got = _run(sess, W, mode, U, I, d, B, u, i, j, calls)
        for x, y, name in zip(ref, got, ("loss", "P", "Q", "mP", "mQ", "vP", "vQ")):
            assert np.array_equal(x, y), "W = %d: %s" % (W, name)


def test_window_is_the_default_where_the_tables_dwarf_the_batch(sess):
    """lazy_window = -1 (the default): 12 batches per launch where both tables hold >= 3 batches' worth of rows and d >= 32 -- seen in
    the launch brackets' unit counts -- and a launch per batch elsewhere; the same bits either way."""
    for (U, I, d, B, nb), windows in (((200_000, 20_000, 64, 1024, 21), True), ((200_000, 5_000, 64, 2048, 9), False),
                                      ((200_000, 20_000, 16, 1024, 9), False), ((200_000, 20_000, 32, 6144, 5), True)):
        u, i, j = _stream(5, U, I, nb * B + 77, 1.1)
        ref = _run(sess, 0, "lazy", U, I, d, B, u, i, j)
        sess.ctx.profile(True)
        try:
            got = _run(sess, -1, "lazy", U, I, d, B, u, i, j)
            launches = sess.ctx.profile_read()["bpr_triples_lazy"][1]
            units = sess.ctx.profile_units("bpr_triples_lazy")
        finally:
            sess.ctx.profile(False)
        assert units > 0 and (units > launches) == windows, (U, I, d, B, units, launches)
        for x, y, name in zip(ref, got, ("loss", "P", "Q", "mP", "mQ", "vP", "vQ")):
            assert np.array_equal(x, y), name


def test_window_rows_beyond_a_buffer_descriptors_reach(sess):
    """User tables of 4.6 GB each (9 M x 128 fp32): rows beyond the 4 GiB a buffer descriptor addresses go through the window's
    64-bit assembly loads / stores, finishers' slots included -- users at the table's end that occur several times per batch and in
    consecutive batches.  Against a launch per batch: bit for bit (compared on the device)."""
    import torch
    from cleverrec_amd import _lib
    if torch.cuda.mem_get_info()[0] < 40 << 30:
        pytest.skip("needs 40 GB of device memory")
    U, I, d, B, nb = 9_000_000, 300_000, 128, 2048, 13
    dev = sess.device
    g = torch.Generator(device=dev).manual_seed(3)
    P0 = torch.empty((U, d), dtype=torch.float32, device=dev).normal_(0, 0.01, generator=g)
    Q0 = torch.empty((I, d), dtype=torch.float32, device=dev).normal_(0, 0.01, generator=g)
    n = nb * B + 99
    u = (U - 1 - torch.randint(0, 700, (n,), generator=g, device=dev)).to(torch.int32)
    u[::3] = torch.randint(0, U, (len(u[::3]),), generator=g, device=dev).to(torch.int32)  # (and cold users anywhere)
    hi, ii, ji = _stream(11, U, I, n, 1.1)
    i, j = sess.to_dev(ii), sess.to_dev(ji)

    def run(W):
        sess.ctx.set_option("lazy_window", W)
        try:
            st = runtime.BprState.__new__(runtime.BprState)
            st.sess = sess
            st.P, st.Q = P0.clone(), Q0.clone()
            st.mP, st.vP, st.mQ, st.vQ = torch.zeros_like(st.P), torch.zeros_like(st.P), torch.zeros_like(st.Q), torch.zeros_like(st.Q)
            st.tables = _lib.BprTables(*(t.data_ptr() for t in (st.P, st.Q, st.mP, st.vP, st.mQ, st.vQ)), U, I, d)
            st.adam = _lib.Adam(1e-3, 0.9, 0.999, 1e-8, 0.01, _lib.ADAM_LAZY, 0.9, 0.999, 0)
            loss = st.train_epoch(u, i, j, B)
            torch.cuda.synchronize()
            return st, loss
        finally:
            sess.ctx.set_option("lazy_window", -1)

    ref, ref_loss = run(0)
    assert float((ref.P[U - 700:] != P0[U - 700:]).float().mean()) > 0.9  # (the end of the table did move)
    got, loss = run(8)
    assert torch.equal(loss, ref_loss)
    for name in ("P", "Q", "mP", "mQ", "vP", "vQ"):
        assert torch.equal(getattr(got, name), getattr(ref, name)), name
