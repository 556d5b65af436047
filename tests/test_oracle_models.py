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
