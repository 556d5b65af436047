"""Row-sharded BPR step (crec_bpr_train_epoch_sharded, BASELINE configs[4] path) on ONE MI355X.

World sizes 2 and 4 run as ranks of an in-process local group (crec_comm_init_local): one ctx, one host thread and
one stream per rank, all on cuda:0 -- the REAL kernels and the REAL exchange sequence (request lists per chunk,
rows forward / gradient rows back per step as all-to-all-v), with device copies standing in for xGMI.  Checked
  * against the oracle's restatement of the sharded arithmetic (per-rank sums, then rank order): tight bounds;
  * 1 rank == 2 ranks == 4 ranks on the same global batches, up to the order of the fp32 additions;
  * world of one against the plain lazy step (crec_bpr_train_epoch) on the same triples.
Index work (compact ids, request lists) is checked bit-exact through its consequences: any wrong row id shows up as
an O(1e-3) error in a table row."""
import threading

import numpy as np
import pytest
import torch

import oracle
from cleverrec_amd import parallel, runtime

pytestmark = pytest.mark.gpu


def _tables(seed, U, I, d, scale=0.01):
    g = np.random.Generator(np.random.PCG64(seed))
    return (g.standard_normal((U, d)) * scale).astype(np.float32), (g.standard_normal((I, d)) * scale).astype(np.float32)


def _streams(seed, U, I, G, n_per_rank, zipf=True):
    """per-rank triples: users from the rank's own range (local ids), positives popular-heavy, negatives uniform"""
    g = np.random.Generator(np.random.PCG64(seed))
    out, ranges = [], []
    for r in range(G):
        lo, hi = parallel.user_range(U, r, G)
        n = n_per_rank[r]
        u = g.integers(0, hi - lo, n).astype(np.int32)
        i = (np.minimum(g.zipf(1.3, n) - 1, I - 1) if zipf else g.integers(0, I, n)).astype(np.int32)
        j = g.integers(0, I, n).astype(np.int32)
        out.append((u, i, j))
        ranges.append((lo, hi))
    return out, ranges


def _run_sharded(G, P0, Q0, ranges, streams, B, d, lr=1e-3, reg=0.01, epochs=1, options=None):
    """G ranks of a local group on cuda:0 -> (P full, Q full, losses [nb], stats of every rank)"""
    I = len(Q0)
    sessions = [runtime.Session(0) for _ in range(G)]
    for s in sessions:
        for k, v in (options or {}).items():
            s.ctx.set_option(k, v)
    if G > 1:
        parallel.init_local_group(sessions)
    states, result, errors = [None] * G, [None] * G, []

    def work(r):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                s = sessions[r]
                lo, hi = ranges[r]
                st = parallel.ShardedBprState(s, P0[lo:hi], parallel.shard_items(Q0, r, G), I, lr, reg, r, G)
                u, i, j = (s.to_dev(x) for x in streams[r])
                losses = [st.train_epoch(u, i, j, B).cpu().numpy() for _ in range(epochs)]
                torch.cuda.current_stream().synchronize()
                states[r] = st
                result[r] = (st.P.cpu().numpy(), st.Q.cpu().numpy(), np.concatenate(losses), st.stats_dict())
        except Exception as e:  # noqa: BLE001
            errors.append((r, e))

    threads = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not any(t.is_alive() for t in threads), "a rank hung"
    if errors:
        raise errors[0][1]
    P = np.concatenate([x[0] for x in result])
    Q = parallel.unshard_items([x[1] for x in result], I)
    for r in range(1, G):
        assert np.array_equal(result[r][2], result[0][2])  # every rank holds the same all-reduced losses
    stats = [x[3] for x in result]
    for s in sessions:
        s.close()
    return P, Q, result[0][2], stats


@pytest.mark.parametrize("G", [1, 2, 4])
@pytest.mark.parametrize("d", [16, 128])
def test_sharded_matches_oracle(G, d):
    U, I, B = 400, 301, 256  # 301: not a multiple of any G, so the last shard rows are padding
    P0, Q0 = _tables(3, U, I, d)
    n = [2 * B + 40, 2 * B + 7, 2 * B + 1, 3 * B][:G]  # ragged last batches, every rank 3 batches
    streams, ranges = _streams(5, U, I, G, n)
    P, Q, loss, stats = _run_sharded(G, P0, Q0, ranges, streams, B, d)
    oP, oQ, oloss, _ = oracle.sharded_steps(P0, Q0, ranges, streams, B)
    np.testing.assert_allclose(loss, oloss, rtol=1e-5)
    np.testing.assert_allclose(P, oP, rtol=0, atol=2e-6)
    np.testing.assert_allclose(Q, oQ, rtol=0, atol=2e-6)
    # what was exchanged: every (batch, unique item) pair once, and served exactly as often as requested
    want = sum(len(np.unique(np.concatenate([i[b * B:(b + 1) * B], j[b * B:(b + 1) * B]]))) for _, i, j in streams for b in range(3))
    assert sum(s["rows_requested"] for s in stats) == want == sum(s["rows_served"] for s in stats)
    if G == 1:
        assert stats[0]["bytes_rows_sent"] == 0 and stats[0]["bytes_grads_sent"] == 0
    else:
        assert sum(s["bytes_rows_sent"] for s in stats) == sum(s["bytes_grads_sent"] for s in stats) > 0


@pytest.mark.parametrize("G", [1, 2, 4])
def test_sharded_through_the_global_memory_index(G):
    """The three index builds of a chunk (plan, owner side, compact-id index) through csrc/index_big.hip -- the form batches beyond
    8192 triples take -- forced onto batches of 256 (`big_index`): the same tables and losses BIT FOR BIT as the in-LDS builds, and
    both against the oracle."""
    U, I, B, d = 400, 301, 256, 16
    P0, Q0 = _tables(3, U, I, d)
    n = [2 * B + 40, 2 * B + 7, 2 * B + 1, 3 * B][:G]
    streams, ranges = _streams(5, U, I, G, n)
    P, Q, loss, stats = _run_sharded(G, P0, Q0, ranges, streams, B, d, options={"big_index": 1})
    P1, Q1, loss1, stats1 = _run_sharded(G, P0, Q0, ranges, streams, B, d)
    assert np.array_equal(P, P1) and np.array_equal(Q, Q1) and np.array_equal(loss, loss1)
    assert [s["rows_requested"] for s in stats] == [s["rows_requested"] for s in stats1]
    oP, oQ, oloss, _ = oracle.sharded_steps(P0, Q0, ranges, streams, B)
    np.testing.assert_allclose(loss, oloss, rtol=1e-5)
    np.testing.assert_allclose(P, oP, rtol=0, atol=2e-6)
    np.testing.assert_allclose(Q, oQ, rtol=0, atol=2e-6)


@pytest.mark.parametrize("G,B", [(2, 12000), (1, 20000)])
def test_sharded_batches_beyond_the_lds_sort(G, B):
    """batch_size > 8192 per rank (what amortises the exchanges' hand-shakes over more triples): ragged last batches, a catalogue
    no world size divides, a Zipf head that one owner serves thousands of times per batch; against the oracle."""
    U, I, d = 3000, 2001, 16
    P0, Q0 = _tables(4, U, I, d)
    n = [2 * B + 333, 2 * B + 5][:G]  # (every rank the same number of batches: the epoch is collective)
    streams, ranges = _streams(6, U, I, G, n)
    P, Q, loss, stats = _run_sharded(G, P0, Q0, ranges, streams, B, d)
    oP, oQ, oloss, _ = oracle.sharded_steps(P0, Q0, ranges, streams, B)
    np.testing.assert_allclose(loss, oloss, rtol=1e-5)
    np.testing.assert_allclose(P, oP, rtol=0, atol=2e-6)
    np.testing.assert_allclose(Q, oQ, rtol=0, atol=2e-6)
    assert sum(s["rows_requested"] for s in stats) == sum(s["rows_served"] for s in stats) > 0


def test_one_two_four_ranks_agree_on_the_same_global_batches():
    """The same global batches (rank-major concatenation of the ranks' batches) on 1, 2 and 4 ranks: same losses,
    same tables up to fp32 add order (a row's gradient is summed per rank first, then over ranks)."""
    U, I, B4, d = 512, 257, 128, 64
    P0, Q0 = _tables(9, U, I, d)
    streams4, ranges4 = _streams(11, U, I, 4, [4 * B4] * 4)
    glob = [(u + lo, i, j) for (u, i, j), (lo, hi) in zip(streams4, ranges4)]

    def merged(G):  # the 4 ranks' batches regrouped for G ranks: rank r of G takes the users of 4/G consecutive quarter-ranks
        per = 4 // G
        out, rng = [], []
        for r in range(G):
            lo, hi = ranges4[r * per][0], ranges4[r * per + per - 1][1]
            parts = glob[r * per:(r + 1) * per]
            u = np.concatenate([np.concatenate([p[0][b * B4:(b + 1) * B4] for p in parts]) for b in range(4)]) - lo
            i = np.concatenate([np.concatenate([p[1][b * B4:(b + 1) * B4] for p in parts]) for b in range(4)])
            j = np.concatenate([np.concatenate([p[2][b * B4:(b + 1) * B4] for p in parts]) for b in range(4)])
            out.append((u.astype(np.int32), i, j))
            rng.append((lo, hi))
        return out, rng

    res = {}
    for G in (1, 2, 4):
        st, rg = merged(G)
        assert rg == [parallel.user_range(U, r, G) for r in range(G)]
        res[G] = _run_sharded(G, P0, Q0, rg, st, B4 * (4 // G), d)
    for G in (2, 4):
        np.testing.assert_allclose(res[G][2], res[1][2], rtol=1e-6)
        for a, b in ((res[G][0], res[1][0]), (res[G][1], res[1][1])):
            diff = np.abs(a - b)
            assert diff.max() <= 2e-5 and (diff > 2e-6).mean() < 1e-3  # a different add order: nearly every element tight


def test_world_of_one_matches_plain_lazy_step():
    """No communicator at all: the sharded epoch (plan, compact table, owner apply) against crec_bpr_train_epoch (lazy)
    on the same triples -- heavy rows included (a 40-item catalogue: every popular item has > 16 occurrences)."""
    U, I, B, d = 300, 40, 1024, 32
    P0, Q0 = _tables(4, U, I, d)
    streams, ranges = _streams(6, U, I, 1, [2 * B + 100])
    P, Q, loss, _ = _run_sharded(1, P0, Q0, ranges, streams, B, d, epochs=2)
    sess = runtime.Session(0)
    st = runtime.BprState(sess, P0, Q0, lr=1e-3, reg=0.01, adam_mode="lazy")
    u, i, j = (sess.to_dev(x) for x in streams[0])
    ref = np.concatenate([st.train_epoch(u, i, j, B).cpu().numpy() for _ in range(2)])
    np.testing.assert_allclose(loss, ref, rtol=1e-6)
    np.testing.assert_allclose(P, st.P.cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(Q, st.Q.cpu().numpy(), rtol=0, atol=2e-6)
    sess.close()


def test_rccl_transport_single_rank_communicator():
    """The RCCL transport itself (crec_comm_init -> ncclCommInitRank, the grouped ncclSend / ncclRecv all-to-all-v of request
    lists, rows and gradient rows) on a communicator of ONE rank -- all one GPU can host: every exchange is a send to self
    through RCCL.  Bit-identical to the same epoch without a communicator."""
    U, I, B, d = 300, 40, 1024, 32
    P0, Q0 = _tables(4, U, I, d)
    streams, ranges = _streams(6, U, I, 1, [2 * B + 100])
    P, Q, loss, _ = _run_sharded(1, P0, Q0, ranges, streams, B, d, epochs=2)
    sess = runtime.Session(0)
    parallel.init_rccl(sess, rank=0, world=1)
    st = parallel.ShardedBprState(sess, P0, parallel.shard_items(Q0, 0, 1), I, 1e-3, 0.01, 0, 1)
    u, i, j = (sess.to_dev(x) for x in streams[0])
    got = np.concatenate([st.train_epoch(u, i, j, B).cpu().numpy() for _ in range(2)])
    torch.cuda.synchronize()
    np.testing.assert_array_equal(got, loss)
    np.testing.assert_array_equal(st.P.cpu().numpy(), P)
    np.testing.assert_array_equal(st.Q.cpu().numpy()[:I], Q)
    assert st.stats_dict()["rows_served"] > 0
    sess.close()


def test_sharded_c5_row_width_and_batch():
    """BASELINE configs[4] row width and batch (d = 128, B = 6144 per rank) on a scaled-down catalogue, 2 ranks."""
    U, I, B, d = 20000, 5003, 6144, 128
    P0, Q0 = _tables(8, U, I, d)
    streams, ranges = _streams(10, U, I, 2, [2 * B, 2 * B])
    P, Q, loss, stats = _run_sharded(2, P0, Q0, ranges, streams, B, d)
    oP, oQ, oloss, _ = oracle.sharded_steps(P0, Q0, ranges, streams, B)
    np.testing.assert_allclose(loss, oloss, rtol=1e-5)
    for a, b in ((P, oP), (Q, oQ)):
        # 2.6 M elements: the odd one whose gradient sum cancels sits an Adam step's fraction away (lr * dg/|g|, see
        # tests/_adam_close.py); all the others are tight
        diff = np.abs(a - b)
        assert diff.max() <= 2e-5 and (diff > 2e-6).mean() <= 1e-5
    assert all(s["max_rows_served_per_batch"] <= 16384 for s in stats)


def test_sharded_eval_matches_reference_loop():
    """parallel.eval_rs_sharded (test users sharded over ranks, one all-reduce of the metric sums) on two user shards run
    one after the other on cuda:0: the combined means equal the reference loop over all users (oracle.test_model_rs_py)."""
    from cleverrec_amd.data import RankingData
    g = np.random.Generator(np.random.PCG64(21))
    U, I, d = 150, 400, 32
    ui_train = {u: [int(x) for x in g.choice(I, size=int(g.integers(2, 25)), replace=False)] for u in range(U) if u % 11 != 10}
    ui_test = {u: [int(x) for x in g.choice(I, size=int(g.integers(1, 4)), replace=False)] for u in range(0, U, 2)}
    data = RankingData(U, I, ui_train, ui_test)
    P, Q = _tables(5, U, I, d, scale=1.0)
    sess = runtime.Session(0)
    tot = np.zeros((3, 2))
    n = 0
    for r in range(2):
        lo, hi = parallel.user_range(U, r, 2)
        shard = parallel.shard_users(data, r, 2)
        hr, mrr, ndcg, cnt = parallel.eval_rs_sharded(sess, sess.to_dev(P[lo:hi]), sess.to_dev(Q), shard, [10, 20], batch_size_t=32)
        tot += np.stack([hr, mrr, ndcg]) * cnt
        n += cnt
    H, M, N = oracle.test_model_rs_py(lambda cur: oracle.scores(P, Q, cur, 1), data.test_users, data.ui_train, data.ui_test, [10, 20], 64)
    assert n == len(data.test_users)
    for kid in range(2):
        np.testing.assert_allclose(tot[0, kid] / n, np.mean(H[kid]), rtol=1e-12)
        np.testing.assert_allclose(tot[1, kid] / n, np.mean(M[kid]), rtol=1e-12)
        np.testing.assert_allclose(tot[2, kid] / n, np.mean(N[kid]), rtol=1e-12)
    sess.close()


def test_an_owner_that_cannot_serve_fails_every_rank_together():
    """What an owner is asked for in a batch is data-dependent and rank-local: here rank 0 owns every requested item (all even ids,
    unique, 2 ranks x 2 x 8192 requests per batch > the 16384 rows its index sorts per batch) while rank 1 is asked for nothing.
    The verdict is collective: BOTH ranks return the error (nobody is left blocked in an exchange)."""
    G, B, d = 2, 8192, 16
    U, I = 64, 2 * 2 * 2 * B + 2
    P0, Q0 = _tables(1, U, I, d)
    ranges = [parallel.user_range(U, r, G) for r in range(G)]
    streams = []
    for r in range(G):
        lo, hi = ranges[r]
        u = (np.arange(B) % (hi - lo)).astype(np.int32)
        i = (2 * (np.arange(B) + r * 2 * B)).astype(np.int32)  # even ids: owner 0; disjoint between the ranks
        j = (2 * (np.arange(B) + r * 2 * B + B)).astype(np.int32)
        streams.append((u, i, j))
    sessions = [runtime.Session(0) for _ in range(G)]
    parallel.init_local_group(sessions)
    errors = [None] * G

    def work(r):
        torch.cuda.set_device(0)
        with torch.cuda.stream(torch.cuda.Stream()):
            s = sessions[r]
            lo, hi = ranges[r]
            st = parallel.ShardedBprState(s, P0[lo:hi], parallel.shard_items(Q0, r, G), I, 1e-3, 0.01, r, G)
            try:
                st.train_epoch(*(s.to_dev(x) for x in streams[r]), B)
                torch.cuda.current_stream().synchronize()
            except Exception as e:  # noqa: BLE001
                errors[r] = str(e)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(G)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not any(t.is_alive() for t in threads), "a rank was left blocked"
    assert errors[0] and "owner-side index" in errors[0]
    assert errors[1] and "peer rank" in errors[1]
    for s in sessions:
        s.close()


def test_sharded_at_a_million_items():
    """BASELINE configs[4]'s catalogue scale on the item side: 1 000 003 items (odd: the last shard row of rank 1 is padding), d = 128,
    B = 6144 per rank, 2 ranks.  The oracle runs the same batches on the compact problem of the touched items (exact: an item no
    batch names takes no part in the row-sharded step); every other row of Q must keep its bits."""
    U, I, B, d, G = 20000, 1_000_003, 6144, 128, 2
    P0, Q0 = _tables(12, U, I, d)
    streams, ranges = _streams(13, U, I, G, [2 * B + 50, 2 * B + 1])  # ragged last batches, both ranks 3 batches (the epoch is collective)
    P, Q, loss, stats = _run_sharded(G, P0, Q0, ranges, streams, B, d)
    qq = np.unique(np.concatenate([np.concatenate([i, j]) for _, i, j in streams]))
    compact = [(u, np.searchsorted(qq, i).astype(np.int32), np.searchsorted(qq, j).astype(np.int32)) for u, i, j in streams]
    oP, oQ, oloss, _ = oracle.sharded_steps(P0, Q0[qq], ranges, compact, B)
    np.testing.assert_allclose(loss, oloss, rtol=1e-5)
    for a, b in ((P, oP), (Q[qq], oQ)):
        diff = np.abs(a - b)  # (see test_sharded_c5_row_width_and_batch: the odd cancelling gradient sits a fraction of an Adam step away)
        assert diff.max() <= 2e-5 and (diff > 2e-6).mean() <= 1e-5
    untouched = np.ones(I, bool)
    untouched[qq] = False
    assert np.array_equal(Q[untouched], Q0[untouched])
    assert sum(s["rows_requested"] for s in stats) == sum(s["rows_served"] for s in stats) > 0


@pytest.mark.parametrize("big", [0, 1])
@pytest.mark.parametrize("G,I,nb", [(1, 53, 64), (2, 53, 64), (4, 1_000_003, 6144), (8, 1_000_003, 8192), (3, 4099, 1), (8, 40, 777),
                                    (4, 1_000_003, 40000)])
def test_index_kernels_equal_the_host_planner(G, I, nb, big):
    """shard_plan_batches + shard_pack_ids and shard_owner_index (what the epoch launches) against csrc/shard_host.cpp (what the
    multi-process gloo test runs as every rank's planner, tests/test_parallel_cpu.py): bit for bit, every output -- by the in-LDS
    builds and (big = 1; a batch of 40 000 always) by the global-memory builds of csrc/index_big.hip."""
    import ctypes as C
    sess = runtime.Session(0)
    sess.ctx.set_option("big_index", big)
    g = np.random.Generator(np.random.PCG64(G * 1000 + nb))
    i = np.minimum(g.zipf(1.2, nb) - 1, I - 1).astype(np.int32)
    j = g.integers(0, I, nb).astype(np.int32)
    keys, off, ci, cj, ids = parallel.plan_batch_host(i, j, G, I)
    d_keys, d_ids = sess.zeros(2 * nb, torch.int32), sess.zeros(2 * nb, torch.int32)
    d_n, d_off = sess.zeros(1, torch.int32), sess.zeros(G + 1, torch.int32)
    d_ci, d_cj = sess.zeros(nb, torch.int32), sess.zeros(nb, torch.int32)
    p = sess.ptr
    d_i, d_j = sess.to_dev(i), sess.to_dev(j)
    sess.ctx.check(sess.lib.crec_debug_shard_plan(sess.ctx.h, p(d_i), p(d_j), nb, G, I, p(d_keys), p(d_n), p(d_off),
                                                  p(d_ci), p(d_cj), p(d_ids), sess.stream_ptr()))
    n = int(d_n.cpu()[0])
    assert n == len(keys)
    assert np.array_equal(d_keys.cpu().numpy()[:n], keys) and np.array_equal(d_off.cpu().numpy(), off)
    assert np.array_equal(d_ci.cpu().numpy(), ci) and np.array_equal(d_cj.cpu().numpy(), cj)
    assert np.array_equal(d_ids.cpu().numpy()[:n], ids)
    # the compact ids name the right rows: keys[ci] is the owner-major key of i
    S = parallel.shard_rows(I, G)
    assert np.array_equal(keys[ci], (i.astype(np.int64) % G) * S + i // G)

    # owner side: every peer's request list is a sorted list of unique rows of this shard
    cnt = g.integers(0, min(S, 2 * nb // G) + 1, G).astype(np.int32)
    recv = np.concatenate([np.sort(g.choice(S, int(c), replace=False)) for c in cnt]).astype(np.int32) if cnt.sum() else np.zeros(0, np.int32)
    order, urows, useg = parallel.owner_index_host(recv, cnt)
    tot = int(cnt.sum())
    d_order, d_urows, d_useg, d_q = (sess.zeros(tot + 64, torch.int32) for _ in range(4))
    d_recv = sess.to_dev(recv) if tot else d_q
    sess.ctx.check(sess.lib.crec_debug_shard_owner_index(sess.ctx.h, p(d_recv), cnt.ctypes.data_as(C.c_void_p), G, S,
                                                         p(d_order), p(d_urows), p(d_useg), p(d_q), sess.stream_ptr()))
    q = int(d_q.cpu()[0])
    assert q == len(urows)
    assert np.array_equal(d_order.cpu().numpy()[:tot], order)
    assert np.array_equal(d_urows.cpu().numpy()[:q], urows) and np.array_equal(d_useg.cpu().numpy()[:q + 1], useg)
    sess.close()
