"""Independent cross-checks of the oracle's TF1-graph restatement (parity unpinned at the
TF boundary, so the restatement is checked against a float64 autograd derivation of
model/ranking/BPR.py:31-44 + the published TF1 Adam update).  CPU only."""
import numpy as np
import pytest
import torch

import oracle


def _toy(seed=0, U=50, I=40, d=16, B=256):
    g = np.random.Generator(np.random.PCG64(seed))
    P = (g.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q = (g.standard_normal((I, d)) * 0.1).astype(np.float32)
    u = g.integers(0, U, B).astype(np.int32)
    i = g.integers(0, I, B).astype(np.int32)
    j = g.integers(0, I, B).astype(np.int32)
    return P, Q, u, i, j


def _autograd_step(P, Q, mP, vP, mQ, vQ, u, i, j, reg, lr, t, mode, b1=0.9, b2=0.999, eps=1e-8):
    """float64 re-derivation: loss of BPR.py:42-43, dense autograd gradient (== summed
    IndexedSlices), TF1 sparse Adam (all rows decay + update in mode 0)."""
    Pt = torch.tensor(P, dtype=torch.float64, requires_grad=True)
    Qt = torch.tensor(Q, dtype=torch.float64, requires_grad=True)
    ut, it, jt = (torch.tensor(x, dtype=torch.long) for x in (u, i, j))
    pu, qi, qj = Pt[ut], Qt[it], Qt[jt]
    x = (pu * qi).sum(1) - (pu * qj).sum(1)
    loss = torch.nn.functional.softplus(-x).sum() + reg * 0.5 * ((pu ** 2).sum() + (qi ** 2).sum() + (qj ** 2).sum())
    loss.backward()
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    out = []
    for var, grad, m, v, idx in ((P, Pt.grad.numpy(), mP, vP, u), (Q, Qt.grad.numpy(), mQ, vQ, np.concatenate([i, j]))):
        touched = np.zeros(len(var), bool)
        touched[idx] = True
        rows = np.ones(len(var), bool) if mode == 0 else touched
        m2, v2, var2 = m.astype(np.float64).copy(), v.astype(np.float64).copy(), var.astype(np.float64).copy()
        m2[rows] = b1 * m2[rows]
        v2[rows] = b2 * v2[rows]
        m2[touched] += (1 - b1) * grad[touched]
        v2[touched] += (1 - b2) * grad[touched] ** 2
        var2[rows] -= lr_t * m2[rows] / (np.sqrt(v2[rows]) + eps)
        out += [var2, m2, v2]
    return float(loss.detach()), out


@pytest.mark.parametrize("mode", [0, 1])
def test_bpr_step_matches_float64_autograd(mode):
    P, Q, u, i, j = _toy()
    m = oracle.BprModel(P, Q, lr=1e-3, reg=0.01, adam_mode=mode)
    ref = [P, np.zeros_like(P), np.zeros_like(P), Q, np.zeros_like(Q), np.zeros_like(Q)]
    for t in range(1, 6):
        loss = m.step(u, i, j)
        rl, (P2, mP2, vP2, Q2, mQ2, vQ2) = _autograd_step(ref[0], ref[3], ref[1], ref[2], ref[4], ref[5], u, i, j,
                                                          0.01, 1e-3, t, mode)
        assert loss == pytest.approx(rl, rel=1e-6)
        # Adam normalises g/sqrt(v): compare the *update* at 1e-4 of lr, the slots relatively
        np.testing.assert_allclose(m.P, P2, rtol=0, atol=2e-7)
        np.testing.assert_allclose(m.Q, Q2, rtol=0, atol=2e-7)
        np.testing.assert_allclose(m.mP, mP2, rtol=2e-5, atol=2e-8)  # cancellation in fp32 row sums
        np.testing.assert_allclose(m.vQ, vQ2, rtol=2e-5, atol=1e-12)
        ref = [a.astype(np.float32) for a in (m.P, m.mP, m.vP, m.Q, m.mQ, m.vQ)]
        u, i, j = np.roll(u, 3), np.roll(i, 5), np.roll(j, 7)


def test_tf1_adam_moves_untouched_rows_lazy_does_not():
    P, Q, u, i, j = _toy(U=60)
    u = (u % 30).astype(np.int32)  # users 30..59 never touched
    a, b = oracle.BprModel(P, Q, adam_mode=0), oracle.BprModel(P, Q, adam_mode=1)
    for _ in range(2):
        a.step(u, i, j)
        b.step(u, i, j)
    assert np.array_equal(b.P[30:], P[30:]) and np.array_equal(a.P[30:], P[30:])  # m == 0 there: no motion either way
    u2 = ((u + 30) % 60).astype(np.int32)
    a.step(u2, i, j)
    b.step(u2, i, j)
    # rows 0..29 were touched before, not now: tf1 keeps decaying m and moving var, lazy freezes them
    assert not np.array_equal(a.P[:30], b.P[:30])
    assert np.all(a.mP[:30] != 0) and np.allclose(a.mP[:30], 0.9 * b.mP[:30], rtol=1e-6)


def test_softplus_restatement():
    z = np.linspace(-30, 30, 2001).astype(np.float32)
    ref = np.logaddexp(0.0, z.astype(np.float64))
    np.testing.assert_allclose(oracle.tf_softplus_np(z), ref, rtol=2e-6, atol=1e-12)


def test_adam_schedule_is_fp32_running_product():
    s = oracle.AdamSchedule(1e-3)
    lr = s.next(2000)
    t = np.arange(1, 2001, dtype=np.float64)
    np.testing.assert_allclose(lr, 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t), rtol=2e-4)
    s2 = oracle.AdamSchedule(1e-3)
    assert np.array_equal(np.concatenate([s2.next(700), s2.next(1300)]), lr)


def test_eval_topk_metrics_match_python_restatement():
    g = np.random.Generator(np.random.PCG64(3))
    U, I, d, K = 37, 90, 8, 20
    P = g.standard_normal((U, d)).astype(np.float32)
    Q = g.standard_normal((I, d)).astype(np.float32)
    Q[5] = Q[4]  # exact ties -> lowest id first
    ui_train = {u: [int(x) for x in g.choice(I, size=int(g.integers(1, 60)), replace=False)] for u in range(U) if u % 5}
    ui_test = {u: [int(x) for x in g.choice(I, size=int(g.integers(1, 6)), replace=False)] for u in range(U) if u % 3}
    users = list(ui_test.keys())
    _, sp, si = oracle.to_csr(ui_train, U)
    _, tp, ti = oracle.to_csr(ui_test, U)
    for order in (0, 1):
        sc = oracle.scores(P, Q, users, order)
        np.testing.assert_allclose(sc, P[users].astype(np.float64) @ Q.T.astype(np.float64), rtol=1e-5, atol=1e-6)
        for asc in (False, True):
            ids, val = oracle.topk(sc, users, sp, si, K, asc)
            hr, mrr, ndcg = oracle.rank_metrics(ids, users, [10, 20], tp, ti)
            H, M, N = oracle.test_model_rs_py(lambda cur: oracle.scores(P, Q, cur, order), users, ui_train, ui_test,
                                              [10, 20], 16, asc)
            for kid in range(2):
                np.testing.assert_allclose(hr[kid], H[kid], rtol=0, atol=0)
                np.testing.assert_allclose(mrr[kid], M[kid], rtol=1e-15)
                np.testing.assert_allclose(ndcg[kid], N[kid], rtol=1e-14)


def test_topk_pads_when_fewer_than_k_unseen():
    sc = np.arange(6, dtype=np.float32)[None, :]
    ptr = np.array([0, 4], dtype=np.int64)
    ids, val = oracle.topk(sc, [0], ptr, np.array([0, 1, 2, 3], np.int32), 4)
    assert ids.tolist() == [[5, 4, -1, -1]]
