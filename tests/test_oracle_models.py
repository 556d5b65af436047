"""The torch restatement's TF1 Adam against the C oracle on the one graph both implement (BPR), and its
two update formulas against a float64 evaluation.  CPU only."""
import numpy as np
import torch

import oracle
from oracle import models_torch as MT


class _BprTorch(MT.Model):
    def __init__(self, P, Q, lr, reg):
        super().__init__({"P": P, "Q": Q}, {"P": "sparse", "Q": "sparse"}, lr)
        self.reg = reg

    def loss(self, w, u, i, j):
        u, i, j = (torch.as_tensor(x, dtype=torch.long) for x in (u, i, j))
        pu, qi, qj = w["P"][u], w["Q"][i], w["Q"][j]
        x = (pu * qi).sum(1) - (pu * qj).sum(1)
        return torch.nn.functional.softplus(-x).sum() + self.reg * (MT._l2(pu) + MT._l2(qi) + MT._l2(qj))


def test_torch_sparse_adam_agrees_with_c_oracle_on_bpr():
    g = np.random.Generator(np.random.PCG64(0))
    U, I, d, B = 40, 30, 8, 128
    P = (g.standard_normal((U, d)) * 0.1).astype(np.float32)
    Q = (g.standard_normal((I, d)) * 0.1).astype(np.float32)
    a, b = _BprTorch(P, Q, 1e-3, 0.01), oracle.BprModel(P, Q, lr=1e-3, reg=0.01, adam_mode=0)
    for s in range(4):
        u, i, j = g.integers(0, U, B), g.integers(0, I, B), g.integers(0, I, B)
        la, lb = a.step(u, i, j), b.step(u, i, j)
        assert abs(la - lb) <= 1e-5 * abs(lb)
    np.testing.assert_allclose(a.numpy("P"), b.P, rtol=0, atol=1e-6)
    np.testing.assert_allclose(a.numpy("Q"), b.Q, rtol=0, atol=1e-6)


def test_dense_formula_is_apply_adam():
    p = {"w": np.array([1.0, -2.0, 0.5], np.float32)}
    opt = MT.TF1Adam({"w": torch.tensor(p["w"])}, {"w": "dense"}, 1e-2)
    gs = [np.array([0.3, -0.1, 0.0], np.float32), np.array([0.2, 0.4, -0.7], np.float32)]
    m = v = np.zeros(3)
    w = p["w"].astype(np.float64)
    for t, g in enumerate(gs, 1):
        opt.step({"w": torch.tensor(g)})
        m = m + (g - m) * 0.1
        v = v + (g * g - v) * 0.001
        w = w - (m * 1e-2 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)) / (np.sqrt(v) + 1e-8)
    np.testing.assert_allclose(opt.p["w"].numpy(), w, rtol=2e-6)


def test_envelope_covers_a_reordered_batch():
    """Model.bounds (the propagated fp32 rounding envelope the GPU tests compare under) must cover what a mere
    change of summation order does: the NeuMF C3-shaped step on a batch and on a permutation of the same batch.
    Also pins that the oracle is deterministic (same batch twice -> bit-identical state)."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _adam_close import assert_adam_close
    g = np.random.Generator(np.random.PCG64(9))
    r = lambda *shape, scale=0.1: (g.standard_normal(shape) * scale).astype(np.float32)
    U, I, B, d = 400, 300, 6144, 64
    Pg, Qg, Pm, Qm = r(U, d), r(I, d), r(U, 64), r(I, 64)
    Ws, bs = [r(128, 64, scale=0.2), r(64, 32, scale=0.2), r(32, 16, scale=0.2)], [r(64, scale=0.05), r(32, scale=0.05), r(16, scale=0.05)]
    h = r(d + 16, scale=0.3)
    mk = lambda: MT.NeuMF(Pg, Qg, Pm, Qm, Ws, bs, h, lr=1e-3, reg1=1e-2, reg2=1e-3)
    a, b, c = mk(), mk(), mk()
    for k in range(3):
        u, i, y = g.integers(0, U, B), g.integers(0, I, B), (g.random(B) < 0.2).astype(np.float32)
        perm = g.permutation(B)
        la, lb, lc = a.step(u, i, y), b.step(u[perm], i[perm], y[perm]), c.step(u, i, y)
        assert la == lc and abs(la - lb) <= 1e-6 * abs(la)
    rep = []
    for name in a.params:
        assert np.array_equal(a.numpy(name), c.numpy(name)), name  # deterministic above torch's 32768-element atomics threshold
        assert_adam_close(a, name, b.params[name], b.opt.m[name], b.opt.v[name], report=rep)
    assert max(x[2] for x in rep) < 0.5  # a reordering alone uses less than half of the envelope
