"""CPU restatement (PyTorch fp32 autograd) of the reference's other hot-path graphs:
GMF / MLP / NeuMF (pointwise, sigmoid cross-entropy), CML, FM -- TEST INFRASTRUCTURE ONLY.

Each forward follows the reference's graph definition line by line (citations per function); the
backward is autograd; the optimizer is a restatement of TF1's AdamOptimizer with its two code paths:

  * 'sparse' (variables reached only through embedding_lookup / gather: IndexedSlices gradients ->
    _apply_sparse_shared):  m = m*b1 + g*(1-b1);  v = v*b2 + g*g*(1-b2);  var -= lr_t*m/(sqrt(v)+eps),
    m, v and var swept on EVERY row (an untouched row is the g = 0 case of the same formula);
  * 'dense' (ApplyAdam kernel):  m += (g-m)*(1-b1);  v += (g*g-v)*(1-b2);  var -= (m*lr_t)/(sqrt(v)+eps).

lr_t = lr*sqrt(1-b2^t)/(1-b1^t) with fp32 running powers.  Parity is unpinned at the TF boundary
(TensorFlow is absent, reference README.md:57); see DESIGN.md 2.

Determinism.  Every embedding_lookup goes through `gather` below, whose backward sums the per-slot gradient
rows serially in slot order -- what TF1's CPU UnsortedSegmentSum does.  (torch's own index backward is
index_put_(accumulate=True), which runs as parallel float atomic adds above 32768 elements: a run-to-run
summation order.)  Model.step also pins torch to one thread so that matmul blocking does not depend on the host.

Rounding-error envelope.  fp32 sums of contribution rows agree between two correct implementations only up to
the order of additions: |delta g| <= NOISE * sum|c_k|.  Adam turns that into lr_t*delta_g/(sqrt(v)+eps) on the
variable -- of the order of lr itself where a gradient sum cancels to ~0 -- so a flat atol on post-Adam weights
is a cliff edge.  Model.step therefore propagates the envelope: tol_m, tol_v, tol_var per element, which
tests/ use as the comparison bound (tight where the gradient is well conditioned, wide only where it cancels).
"""
import numpy as np
import torch

f32 = torch.float32
NOISE = 2e-5  # relative to sum|c_k|: contributions come out of length-d dot products (and a tower) rounded in another
              # order; a pure permutation of the batch already uses 0.6 of a 4e-6 envelope (tests/test_oracle_models.py)


class _Gather(torch.autograd.Function):
    """table[idx] (tf.nn.embedding_lookup / tf.gather) with a serial, slot-ordered backward; also leaves
    sum_k |c_k| per table element in sink[key] (the rounding envelope's scale)."""

    @staticmethod
    def forward(ctx, table, idx, sink, key):
        ctx.save_for_backward(idx)
        ctx.shape, ctx.sink, ctx.key = tuple(table.shape), sink, key
        return table.detach()[idx]

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        flat = idx.numpy().reshape(-1)
        rows = np.ascontiguousarray(g.numpy(), dtype=np.float32).reshape((flat.shape[0],) + ctx.shape[1:])
        out = np.zeros(ctx.shape, np.float32)
        np.add.at(out, flat, rows)  # unbuffered, in slot order: one fixed fp32 summation order
        if ctx.sink is not None:
            a = np.zeros(ctx.shape, np.float32)
            np.add.at(a, flat, np.abs(rows))
            ctx.sink[ctx.key] = ctx.sink.get(ctx.key, 0) + a
            named = ctx.sink.setdefault("_touched", {}).setdefault(ctx.key, np.zeros(ctx.shape[0], bool))
            named[flat] = True  # the rows this lookup names (LazyAdam)
        return torch.from_numpy(out), None, None, None


def gather(w, key, idx):
    """w: the step's leaf dict (may carry '_abs', the |c| sink); key: variable name; idx: long tensor."""
    return _Gather.apply(w[key], idx, w.get("_abs"), key)


class TF1Adam:
    def __init__(self, params, kinds, lr, beta1=0.9, beta2=0.999, eps=1e-8, optimizer="Adam"):
        """params: dict name -> tensor (updated in place); kinds: dict name -> 'sparse' | 'dense'.
        optimizer (get_optimizer, utils/tools.py:79-87): 'Adam' | 'SGD' (GradientDescentOptimizer: var -= lr*g; an
        IndexedSlices gradient is scatter_sub on its unique rows, i.e. the same formula with g = 0 elsewhere) | 'Adagrad'
        (AdagradOptimizer, initial_accumulator_value 0.1 kept in m: accum += g*g; var -= lr*g/sqrt(accum), again g = 0 on
        the rows a sparse gradient does not name)."""
        self.lazy = optimizer == "LazyAdam"  # Adam's sparse formula on the rows a step's gathers name; other rows and their slots stay
        optimizer = "Adam" if self.lazy else optimizer
        self.p, self.kinds, self.optimizer = params, kinds, optimizer
        assert optimizer in ("Adam", "SGD", "Adagrad")
        self.lr, self.b1, self.b2, self.eps = (np.float32(x) for x in (lr, beta1, beta2, eps))
        self.b1p, self.b2p = np.float32(beta1), np.float32(beta2)
        self.m = {k: torch.full_like(v, 0.1) if optimizer == "Adagrad" else torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        # rounding-error envelopes (see the module docstring), float64
        self.tol_m = {k: np.zeros(tuple(v.shape)) for k, v in params.items()}
        self.tol_v = {k: np.zeros(tuple(v.shape)) for k, v in params.items()}
        self.tol_var = {k: np.zeros(tuple(v.shape)) for k, v in params.items()}

    def _envelope(self, k, g, gabs, lr_t):
        """after the update of variable k with gradient g: |delta g| <= NOISE * (sum|c| of the gathers + |g|)
        (dense gradients: NOISE * max|g| as the scale of the cancelled terms)."""
        ga = np.abs(g.numpy().astype(np.float64))
        dg = NOISE * (np.asarray(gabs, dtype=np.float64) + ga) if gabs is not None else NOISE * np.maximum(ga, ga.max(initial=0.0))
        if self.optimizer == "SGD":
            self.tol_var[k] = self.tol_var[k] + float(self.lr) * dg
            return
        if self.optimizer == "Adagrad":  # accum (in m) += g^2; var -= lr*g/sqrt(accum): first-order in dg and in the accumulator's own error
            acc = self.m[k].numpy().astype(np.float64)
            self.tol_m[k] = self.tol_m[k] + 2 * ga * dg + dg * dg
            self.tol_var[k] = self.tol_var[k] + float(self.lr) * (dg / np.sqrt(acc) + 0.5 * ga * self.tol_m[k] / acc ** 1.5) + 2e-7 * float(self.lr) * ga / np.sqrt(acc)
            return
        b1, b2 = float(self.b1), float(self.b2)
        self.tol_m[k] = b1 * self.tol_m[k] + (1 - b1) * dg
        self.tol_v[k] = b2 * self.tol_v[k] + (1 - b2) * (2 * ga * dg + dg * dg)
        # a change of this step's g moves var by lr_t*dg/(sqrt(v)+eps) over the following steps (the (1-b1)*b1^k
        # weights of m sum to one); never by more than a few whole steps
        sens = float(lr_t) * dg / (np.sqrt(self.v[k].numpy().astype(np.float64)) + float(self.eps))
        self.tol_var[k] = self.tol_var[k] + np.minimum(sens, 4.0 * float(self.lr))

    def step(self, grads, gabs=None):
        lr_t = np.float32(np.float32(self.lr * np.sqrt(np.float32(1) - self.b2p)) / (np.float32(1) - self.b1p))
        omb1, omb2 = np.float32(1) - self.b1, np.float32(1) - self.b2
        touched = (gabs or {}).get("_touched", {})
        for k, g in grads.items():
            if g is None:
                continue  # minimize() skips variables without a gradient
            var, m, v = self.p[k], self.m[k], self.v[k]
            keep = None
            if self.lazy and self.kinds[k] == "sparse":  # everything of the untouched rows is put back after the sweep below
                rows = torch.from_numpy(~touched[k])
                keep = (rows, var[rows].clone(), m[rows].clone(), v[rows].clone(),
                        [t[k][rows.numpy()].copy() for t in (self.tol_var, self.tol_m, self.tol_v)])
            if self.optimizer == "SGD":
                var.sub_(float(self.lr) * g)
            elif self.optimizer == "Adagrad":
                m.add_(g * g)
                var.sub_((float(self.lr) * g) / m.sqrt())
            elif self.kinds[k] == "sparse":
                m.mul_(float(self.b1)).add_(g * float(omb1))
                v.mul_(float(self.b2)).add_((g * g) * float(omb2))
                var.sub_((float(lr_t) * m) / (v.sqrt() + float(self.eps)))
            else:
                m.add_((g - m) * float(omb1))
                v.add_((g * g - v) * float(omb2))
                var.sub_((m * float(lr_t)) / (v.sqrt() + float(self.eps)))
            self._envelope(k, g, (gabs or {}).get(k), lr_t)
            if keep is not None:
                rows, var0, m0, v0, tols = keep
                var[rows], m[rows], v[rows] = var0, m0, v0
                for t, t0 in zip((self.tol_var, self.tol_m, self.tol_v), tols):
                    t[k][rows.numpy()] = t0
        self.b1p = np.float32(self.b1p * self.b1)
        self.b2p = np.float32(self.b2p * self.b2)
        return float(lr_t)


def _l2(t):  # tf.nn.l2_loss
    return (t * t).sum() / 2


def _ce_sum(labels, logits):  # utils/tools.py:68-69: reduce_sum(sigmoid_cross_entropy_with_logits)
    return (torch.clamp(logits, min=0) - logits * labels + torch.log1p(torch.exp(-logits.abs()))).sum()


def _pointwise_loss(loss_func, labels, logits):  # get_loss(self.loss_func, self.y, self.logits), utils/tools.py:66-76
    if loss_func == "square":
        return ((labels - logits) ** 2).sum()  # reduce_sum(squared_difference(y, logits)), :74-75
    assert loss_func == "cross_entropy"
    return _ce_sum(labels, logits)


class Model:
    """params (fp32 tensors), kinds, loss(batch) -> scalar.  step(batch) runs one sess.run([train, loss])."""

    def __init__(self, params, kinds, lr, optimizer="Adam"):
        self.params = {k: torch.tensor(np.array(v, dtype=np.float32)) for k, v in params.items()}
        self.opt = TF1Adam(self.params, kinds, lr, optimizer=optimizer)

    def step(self, *batch):
        nt = torch.get_num_threads()
        torch.set_num_threads(1)  # matmul / reduction blocking independent of the host's core count
        try:
            leaves = {k: v.clone().requires_grad_(True) for k, v in self.params.items()}
            w = dict(leaves)
            w["_abs"] = {}
            loss = self.loss(w, *batch)
            grads = torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)
            self.opt.step(dict(zip(leaves.keys(), grads)), w["_abs"])
        finally:
            torch.set_num_threads(nt)
        return float(loss.detach())

    def loss64(self, *batch):
        """the same graph evaluated in float64 on the current parameters (no update): the exact value any fp32
        implementation of the loss is an approximation of."""
        w = {k: v.double() for k, v in self.params.items()}
        self._f64 = True
        try:
            return float(self.loss(w, *batch))
        finally:
            self._f64 = False

    def _float(self, a):  # labels / values in the dtype of the current evaluation
        return torch.as_tensor(np.asarray(a, dtype=np.float64 if getattr(self, "_f64", False) else np.float32))

    def bounds(self, k):
        """(tol_var, tol_m, tol_v) of variable k, float64 arrays: the propagated rounding envelope."""
        return self.opt.tol_var[k], self.opt.tol_m[k], self.opt.tol_v[k]

    def numpy(self, k):
        return self.params[k].numpy()


class BPR(Model):
    """model/ranking/BPR.py:31-44: x = <p_u, q_i> - <p_u, q_j>, loss = -sum(log_sigmoid(x)) + reg * (l2(p_u) + l2(q_i) + l2(q_j)) over
    the batch with duplicates; both tables reached by gathers (IndexedSlices Adam, swept over every row).  The C oracle
    (oracle/crec_oracle.c) restates the same step; this one carries the rounding envelope (bounds())."""

    def __init__(self, P, Q, lr, reg, optimizer="Adam"):
        super().__init__({"P": P, "Q": Q}, {"P": "sparse", "Q": "sparse"}, lr, optimizer)
        self.reg = reg

    def loss(self, w, u, i, j):
        u, i, j = (torch.as_tensor(np.asarray(x), dtype=torch.long) for x in (u, i, j))
        pu, qi, qj = gather(w, "P", u), gather(w, "Q", i), gather(w, "Q", j)
        x = (pu * qi).sum(1) - (pu * qj).sum(1)
        return torch.nn.functional.softplus(-x).sum() + self.reg * (_l2(pu) + _l2(qi) + _l2(qj))

    def loss_slack(self, *batch):
        """how far the loss of this batch may move when every parameter moves inside its rounding envelope (first order):
        sum |dL/dw| * tol_var -- what two correct fp32 trajectories may differ by after many steps, seen from the loss."""
        leaves = {k: v.clone().requires_grad_(True) for k, v in self.params.items()}
        g = torch.autograd.grad(self.loss(dict(leaves), *batch), list(leaves.values()), allow_unused=True)
        return float(sum((gk.abs().double().numpy() * self.opt.tol_var[k]).sum() for k, gk in zip(leaves, g) if gk is not None))


class GMF(Model):
    """model/ranking/GMF.py:29-57: logit = sum_d pu*qi*h (:42); loss = sum CE + reg*(l2(pu)+l2(qi)) (:48)."""

    def __init__(self, P, Q, h, lr, reg, loss_func="cross_entropy", optimizer="Adam"):
        super().__init__({"P": P, "Q": Q, "h_gmf": h}, {"P": "sparse", "Q": "sparse", "h_gmf": "dense"}, lr, optimizer)
        self.reg, self.loss_func = reg, loss_func

    def loss(self, w, u, i, y):
        u, i = torch.as_tensor(u, dtype=torch.long), torch.as_tensor(i, dtype=torch.long)
        y = self._float(y)
        pu, qi = gather(w, "P", u), gather(w, "Q", i)
        logits = ((pu * qi) * w["h_gmf"]).sum(1)
        return _pointwise_loss(self.loss_func, y, logits) + self.reg * (_l2(pu) + _l2(qi))

    def predict(self, users):  # GMF.py:40,57: sigmoid((pu * Q) . h) for every item
        P, Q, h = (self.params[k] for k in ("P", "Q", "h_gmf"))
        return torch.sigmoid((P[torch.as_tensor(users, dtype=torch.long)] * h) @ Q.T).numpy()


def _tower(y, w, n_layers, prefix=""):
    for k in range(n_layers):  # MLP.py:49-53: relu(y W_k + b_k) for every layer
        y = torch.relu(y @ w[prefix + "W_%d" % k] + w[prefix + "b_%d" % k])
    return y


class MLP(Model):
    """model/ranking/MLP.py:31-77: P,Q of width layers[0]//2, W_k [layers[k], layers[k]//2], b_k, h_mlp;
    logit = tower(concat(pu, qi)) . h_mlp (:59); loss = sum CE + reg*(l2(pu)+l2(qi)) (:66)."""

    def __init__(self, P, Q, Ws, bs, h, lr, reg, loss_func="cross_entropy", optimizer="Adam"):
        self.loss_func = loss_func
        params, kinds = {"P": P, "Q": Q, "h_mlp": h}, {"P": "sparse", "Q": "sparse", "h_mlp": "dense"}
        for k, (W, b) in enumerate(zip(Ws, bs)):
            params["W_%d" % k], params["b_%d" % k] = W, b
            kinds["W_%d" % k] = kinds["b_%d" % k] = "dense"
        super().__init__(params, kinds, lr, optimizer)
        self.reg, self.n_layers = reg, len(Ws)

    def loss(self, w, u, i, y):
        u, i = torch.as_tensor(u, dtype=torch.long), torch.as_tensor(i, dtype=torch.long)
        y = self._float(y)
        pu, qi = gather(w, "P", u), gather(w, "Q", i)
        logits = _tower(torch.cat([pu, qi], 1), w, self.n_layers) @ w["h_mlp"]
        return _pointwise_loss(self.loss_func, y, logits) + self.reg * (_l2(pu) + _l2(qi))

    def predict(self, users):  # MLP.py:45-52,71-77 for every item
        w = self.params
        out = []
        for uu in torch.as_tensor(users, dtype=torch.long):
            x = torch.cat([w["P"][uu].expand(len(w["Q"]), -1), w["Q"]], 1)
            out.append(torch.sigmoid(_tower(x, w, self.n_layers) @ w["h_mlp"]))
        return torch.stack(out).numpy()


class NeuMF(Model):
    """model/ranking/NeuMF.py:33-105: four tables; logit = concat(pg*qg, tower(concat(pm, qm))) . h_neumf (:84);
    loss = sum CE + reg1*(l2(pg)+l2(qg)) + reg2*(l2(pm)+l2(qm)) (:93-94).  h_gmf / h_mlp exist as variables but
    receive no gradient."""

    def __init__(self, Pg, Qg, Pm, Qm, Ws, bs, h, lr, reg1, reg2, loss_func="cross_entropy", optimizer="Adam"):
        self.loss_func = loss_func
        params = {"P_gmf": Pg, "Q_gmf": Qg, "P_mlp": Pm, "Q_mlp": Qm, "h_neumf": h}
        kinds = {"P_gmf": "sparse", "Q_gmf": "sparse", "P_mlp": "sparse", "Q_mlp": "sparse", "h_neumf": "dense"}
        for k, (W, b) in enumerate(zip(Ws, bs)):
            params["W_%d" % k], params["b_%d" % k] = W, b
            kinds["W_%d" % k] = kinds["b_%d" % k] = "dense"
        super().__init__(params, kinds, lr, optimizer)
        self.reg1, self.reg2, self.n_layers = reg1, reg2, len(Ws)

    def loss(self, w, u, i, y):
        u, i = torch.as_tensor(u, dtype=torch.long), torch.as_tensor(i, dtype=torch.long)
        y = self._float(y)
        pg, qg, pm, qm = gather(w, "P_gmf", u), gather(w, "Q_gmf", i), gather(w, "P_mlp", u), gather(w, "Q_mlp", i)
        y_mlp = _tower(torch.cat([pm, qm], 1), w, self.n_layers)
        logits = torch.cat([pg * qg, y_mlp], 1) @ w["h_neumf"]
        return _pointwise_loss(self.loss_func, y, logits) + self.reg1 * (_l2(pg) + _l2(qg)) + self.reg2 * (_l2(pm) + _l2(qm))

    def predict(self, users):  # NeuMF.py:66,73,84,105 for every item
        w = self.params
        users = torch.as_tensor(users, dtype=torch.long)
        out = []
        for uu in users:
            pg, pm = w["P_gmf"][uu], w["P_mlp"][uu]
            y_mlp = _tower(torch.cat([pm.expand(len(w["Q_mlp"]), -1), w["Q_mlp"]], 1), w, self.n_layers)
            out.append(torch.sigmoid(torch.cat([pg * w["Q_gmf"], y_mlp], 1) @ w["h_neumf"]))
        return torch.stack(out).numpy()


class CML(Model):
    """model/ranking/CML.py:34-84.  d_ui = |pu-qi|^2 (:42), d_un[b,k] = |pu-q_neg|^2 (:43),
    loss_b = max(d_ui + margin - min_k d_un, 0) * log(rank+1), rank = mean_k(impostor)*I/neg_ratio without
    gradient (:45-53), + reg * sum offdiag(cov([Q;P])) (:63-70).  Both tables get dense gradients (the
    covariance term reads the full tables) -> dense Adam."""

    def __init__(self, P, Q, lr, reg, margin, item_nums, optimizer="Adam"):
        super().__init__({"P": P, "Q": Q}, {"P": "dense", "Q": "dense"}, lr, optimizer)
        self.reg, self.margin, self.item_nums = reg, margin, item_nums

    def loss(self, w, u, i, neg):
        u, i = torch.as_tensor(u, dtype=torch.long), torch.as_tensor(i, dtype=torch.long)
        neg = torch.as_tensor(np.asarray(neg), dtype=torch.long)
        pu, qi, qn = gather(w, "P", u), gather(w, "Q", i), gather(w, "Q", neg)  # qn [B, K, d]
        d_ui = ((pu - qi) ** 2).sum(1)
        d_un = ((pu[:, None, :] - qn) ** 2).sum(2)
        base = torch.clamp(d_ui + self.margin - d_un.min(1).values, min=0)
        imp = ((d_ui[:, None] + self.margin - d_un) > 0).to(d_ui.dtype)
        rank = imp.mean(1) * self.item_nums / neg.shape[1]
        per_pair = base * torch.log(rank + 1).detach()
        X = torch.cat([w["Q"], w["P"]], 0)
        n_rows = X.shape[0]
        Xc = X - X.mean(0)
        cov = (Xc.T @ Xc) / n_rows
        cov_loss = self.reg * (cov.sum() - torch.diagonal(cov).sum())
        return per_pair.sum() + cov_loss

    def predict(self, users):  # CML.py:76,84: |clip_1(pu) - Q_i|^2, smaller is better
        P, Q = self.params["P"], self.params["Q"]
        pu = P[torch.as_tensor(users, dtype=torch.long)]
        nrm = pu.norm(dim=1, keepdim=True)
        pu = torch.where(nrm > 1.0, pu / nrm, pu)  # tf.clip_by_norm(.., 1.0, axes=[1])
        return ((pu[:, None, :] - Q[None]) ** 2).sum(2).numpy()

    def pair_scores(self, u, items):  # CML.py:42,81-82 (loo / sampled negatives): u_i_dist = |pu - qi|^2 with the UNCLIPPED pu
        P, Q = self.params["P"], self.params["Q"]
        return ((P[int(u)][None, :] - Q[torch.as_tensor(np.asarray(items), dtype=torch.long)]) ** 2).sum(1).numpy()


class SBPR(Model):
    """model/ranking/SBPR.py:31-55: x = p.q + bias; loss = sum -log sigmoid((x_ui - x_uk)/suk) + sum -log sigmoid(x_uk - x_uj)
    + reg*(l2 of the four gathered rows and the three gathered biases).  All three variables are reached through tf.gather:
    IndexedSlices Adam (every row swept)."""

    def __init__(self, P, Q, bias, lr, reg, optimizer="Adam"):
        super().__init__({"P": P, "Q": Q, "bias": bias}, {"P": "sparse", "Q": "sparse", "bias": "sparse"}, lr, optimizer)
        self.reg = reg

    def loss(self, w, u, i, k, j, suk):
        u, i, k, j = (torch.as_tensor(np.asarray(x), dtype=torch.long) for x in (u, i, k, j))
        suk = self._float(suk)
        pu = gather(w, "P", u)
        qi, qk, qj = gather(w, "Q", i), gather(w, "Q", k), gather(w, "Q", j)
        bi, bk, bj = gather(w, "bias", i), gather(w, "bias", k), gather(w, "bias", j)
        xi, xk, xj = (pu * qi).sum(1) + bi, (pu * qk).sum(1) + bk, (pu * qj).sum(1) + bj
        sp = torch.nn.functional.softplus
        return sp(-(xi - xk) / suk).sum() + sp(-(xk - xj)).sum() + \
            self.reg * (_l2(pu) + _l2(qi) + _l2(qk) + _l2(qj) + _l2(bi) + _l2(bk) + _l2(bj))


class CUNE_BPR(Model):
    """model/ranking/CUNE_BPR.py:44-61: SBPR's three biased scores; loss = sum -log sigmoid(x_ui - x_uk) + sum -log sigmoid((x_uk -
    x_uj) / (s + 1.0)) + reg*(l2 of the four gathered rows and the three gathered biases).  `s` (tf.Variable(0.0), :42) is a
    trainable scalar used directly: dense ApplyAdam; P, Q, bias through tf.gather: IndexedSlices Adam."""

    def __init__(self, P, Q, bias, s, lr, reg, optimizer="Adam"):
        super().__init__({"P": P, "Q": Q, "bias": bias, "s": np.asarray(s, np.float32).reshape(1)},
                         {"P": "sparse", "Q": "sparse", "bias": "sparse", "s": "dense"}, lr, optimizer)
        self.reg = reg

    def loss(self, w, u, i, k, j):
        u, i, k, j = (torch.as_tensor(np.asarray(x), dtype=torch.long) for x in (u, i, k, j))
        pu = gather(w, "P", u)
        qi, qk, qj = gather(w, "Q", i), gather(w, "Q", k), gather(w, "Q", j)
        bi, bk, bj = gather(w, "bias", i), gather(w, "bias", k), gather(w, "bias", j)
        xi, xk, xj = (pu * qi).sum(1) + bi, (pu * qk).sum(1) + bk, (pu * qj).sum(1) + bj
        sp = torch.nn.functional.softplus
        if w.get("_abs") is not None:  # s sums B terms of either sign (Model.bounds): the rounding scale is sum|ds_r|
            with torch.no_grad():
                b = (xk - xj) / (w["s"] + 1.0)
                w["_abs"]["s"] = w["_abs"].get("s", 0) + (torch.sigmoid(-b) * b.abs() / (w["s"] + 1.0).abs()).sum().reshape(1)
        return sp(-(xi - xk)).sum() + sp(-(xk - xj) / (w["s"] + 1.0)).sum() + \
            self.reg * (_l2(pu) + _l2(qi) + _l2(qk) + _l2(qj) + _l2(bi) + _l2(bk) + _l2(bj))


class LRML(Model):
    """model/ranking/LRML.py:36-63: joint = p_u * q; a = softmax(joint K); r = a M; dist = |p_u + r - q|^2;
    loss = sum max(dist_i - dist_j + margin, 0) (utils/tools.py:72-73) + reg*(l2(p_u) + l2(q_i) + l2(q_j)).
    P, Q are reached through tf.gather (IndexedSlices Adam), K and M are dense variables (ApplyAdam).  _unit_clipping
    (LRML.py:65-68) is never called and the variables' `regularizer=` never reaches the loss."""

    def __init__(self, P, Q, K, M, lr, reg, margin, optimizer="Adam"):
        super().__init__({"P": P, "Q": Q, "K": K, "M": M}, {"P": "sparse", "Q": "sparse", "K": "dense", "M": "dense"}, lr, optimizer)
        self.reg, self.margin = reg, margin

    @staticmethod
    def _lram(w, pu, q):  # LRML.py:43-52 (training form)
        att = torch.softmax((pu * q) @ w["K"], dim=1)
        return att @ w["M"]

    def loss(self, w, u, i, j):
        u, i, j = (torch.as_tensor(np.asarray(x), dtype=torch.long) for x in (u, i, j))
        pu, qi, qj = gather(w, "P", u), gather(w, "Q", i), gather(w, "Q", j)
        fi, fj = pu + self._lram(w, pu, qi) - qi, pu + self._lram(w, pu, qj) - qj
        di, dj = (fi ** 2).sum(1), (fj ** 2).sum(1)
        if w.get("_abs") is not None:
            # rounding scale beyond sum|c_k|: a slot's gradient holds the direct terms +-2g*diff -- for p_u the DIFFERENCE
            # 2g*(diff_i - diff_j), in which p_u itself cancels -- and what comes back through the memory module, where
            # the softmax backward a*(dA - <a, dA>) cancels again and K, M mix all d components of 2g*diff into each
            # one.  Every term is rounded at its own magnitude (~8 ulp: 0.025 * NOISE of it, as in FM below), and that
            # magnitude is the row's largest |2g*diff| component, not the element's own
            g2 = 2.0 * (di - dj + self.margin >= 0).double().numpy()[:, None]
            ai = 0.025 * g2 * np.abs(fi.detach().double().numpy()).max(1, keepdims=True) * np.ones((1, fi.shape[1]))
            aj = 0.025 * g2 * np.abs(fj.detach().double().numpy()).max(1, keepdims=True) * np.ones((1, fj.shape[1]))
            a = np.zeros(tuple(w["P"].shape), np.float64)
            np.add.at(a, u.numpy(), ai + aj)
            w["_abs"]["P"] = w["_abs"].get("P", 0) + a
            a = np.zeros(tuple(w["Q"].shape), np.float64)
            np.add.at(a, i.numpy(), ai)
            np.add.at(a, j.numpy(), aj)
            w["_abs"]["Q"] = w["_abs"].get("Q", 0) + a
        return torch.clamp(di - dj + self.margin, min=0).sum() + self.reg * (_l2(pu) + _l2(qi) + _l2(qj))

    def scores(self, users):
        """pre_scores of the rs branch (LRML.py:74-76): [len(users), I] squared distances, float64."""
        w = {k: v.double() for k, v in self.params.items()}
        pu = w["P"][torch.as_tensor(np.asarray(users), dtype=torch.long)]
        joint = pu[:, None, :] * w["Q"][None, :, :]
        r = torch.softmax(joint @ w["K"], dim=2) @ w["M"]
        return ((pu[:, None, :] + r - w["Q"][None, :, :]) ** 2).sum(2).numpy()


class TransCF(Model):
    """model/ranking/TransCF.py:39-77: NU = A_ui Q, NI = A_iu P (utils/tools.py:100-113); dist = |p_u + NU[u]*NI[i] - q_i|^2;
    loss = sum max(x + margin, 0) + reg1*(|p_u - NU[u]|^2 + |q_i - NI[i]|^2) + reg2*sum (x + margin)^2, x = dist_i - dist_j.
    Both tables receive dense gradients (the sparse-dense products), hence the dense ApplyAdam kernel."""

    def __init__(self, P, Q, ui_train, lr, reg1, reg2, margin, optimizer="Adam"):
        super().__init__({"P": P, "Q": Q}, {"P": "dense", "Q": "dense"}, lr, optimizer)
        self.reg1, self.reg2, self.margin = reg1, reg2, margin
        U, I = np.asarray(P).shape[0], np.asarray(Q).shape[0]
        ui, iu, cnt = [], [], {}
        for u, items in ui_train.items():
            for it in items:
                ui.append((u, it, np.float32(1.0 / len(items))))
                cnt[it] = cnt.get(it, 0) + 1
        for u, it, _ in ui:
            iu.append((it, u, np.float32(1.0 / cnt[it])))

        def sp(trip, shape):
            idx = torch.tensor([[a for a, _, _ in trip], [b for _, b, _ in trip]], dtype=torch.long)
            return idx, torch.tensor([float(c) for _, _, c in trip], dtype=f32), shape
        self._ui, self._iu = sp(ui, (U, I)), sp(iu, (I, U))

    def _nbr(self, w):
        def mm(spec, dense):
            idx, val, shape = spec
            return torch.sparse.mm(torch.sparse_coo_tensor(idx, val.to(dense.dtype), shape), dense)
        return mm(self._ui, w["Q"]), mm(self._iu, w["P"])

    def loss(self, w, u, i, j):
        u, i, j = (torch.as_tensor(np.asarray(x), dtype=torch.long) for x in (u, i, j))
        NU, NI = self._nbr(w)
        pu, qi, qj = w["P"][u], w["Q"][i], w["Q"][j]
        nu, ni, nj = NU[u], NI[i], NI[j]
        di = ((pu + nu * ni - qi) ** 2).sum(1)
        dj = ((pu + nu * nj - qj) ** 2).sum(1)
        reg_nbr = ((pu - nu) ** 2).sum() + ((qi - ni) ** 2).sum()
        reg_dist = ((di + self.margin - dj) ** 2).sum()
        return torch.clamp(di - dj + self.margin, min=0).sum() + (self.reg1 * reg_nbr + self.reg2 * reg_dist)

    def scores(self, users, clip_users=True):
        """pre_scores (TransCF.py:85-91): rs branch with clip_1(p_u), loo branch with p_u; [len(users), I], float64."""
        w = {k: v.double() for k, v in self.params.items()}
        NU, NI = self._nbr(w)
        uu = torch.as_tensor(np.asarray(users), dtype=torch.long)
        a = w["P"][uu]
        if clip_users:
            a = a / torch.clamp(a.norm(dim=1, keepdim=True), min=1.0)
        return ((a[:, None, :] + NU[uu][:, None, :] * NI[None, :, :] - w["Q"][None, :, :]) ** 2).sum(2).numpy()


class FM(Model):
    """model/rating/FM.py:20-53: y = w0 + sum_f wi[x_f] + 0.5*sum_k[(sum_f x v)^2 - sum_f x^2 v^2] (the linear
    term ignores x_value, :35,49); loss = sum (y - y_pre)^2 + reg*(l2(wi) + l2(vif)) over the FULL tables (:52)
    -> dense Adam for wi and vif; w0 dense."""

    def __init__(self, w0, wi, vif, lr, reg, optimizer="Adam"):
        super().__init__({"w0": w0, "wi": wi, "vif": vif}, {"w0": "dense", "wi": "dense", "vif": "dense"}, lr, optimizer)
        self.reg = reg

    def y_pre(self, w, x_idx, x_val):
        x_idx = torch.as_tensor(np.asarray(x_idx), dtype=torch.long)
        x_val = self._float(x_val)
        wi_e, v_e = gather(w, "wi", x_idx), gather(w, "vif", x_idx)  # [B,F], [B,F,d]
        sq_sum = ((x_val[:, :, None] * v_e).sum(1)) ** 2
        sum_sq = ((x_val ** 2)[:, :, None] * (v_e ** 2)).sum(1)
        return w["w0"] + wi_e.sum(1) + 0.5 * (sq_sum - sum_sq).sum(1)

    def loss(self, w, x_idx, x_val, y):
        y = self._float(y)
        yp = self.y_pre(w, x_idx, x_val)
        self.last_y_pre = yp.detach().numpy()
        if w.get("_abs") is not None:
            # rounding scale of d loss / d vif beyond sum|c_k|: the slot gradient -2(y-y')*x_f*(S - x_f*v_f), S = sum_f x_f*v_f,
            # is a DIFFERENCE of two terms each rounded at its own magnitude (the sum / sum-of-squares trick, FM.py:43-49;
            # autograd: x_f*S and -x_f^2*v_f), so its rounding error scales with |x_f|*(|S| + |x_f*v_f|), not with the result
            xi = np.asarray(x_idx).reshape(-1)
            xv = np.asarray(x_val, dtype=np.float64)
            ve = w["vif"].detach().numpy().astype(np.float64)[np.asarray(x_idx)]  # [B,F,d]
            xve = xv[:, :, None] * ve
            S = xve.sum(1)
            r = 2.0 * np.abs(np.asarray(y, dtype=np.float64) - yp.detach().numpy().astype(np.float64))
            # (plain fp32 rounding of the two terms, ~8 ulp: 5e-7, i.e. 0.025 * NOISE of their magnitude)
            term = 0.025 * r[:, None, None] * np.abs(xv)[:, :, None] * (np.abs(S)[:, None, :] + np.abs(xve))
            a = np.zeros(tuple(w["vif"].shape), np.float64)
            np.add.at(a, xi, term.reshape(len(xi), -1))
            w["_abs"]["vif"] = w["_abs"].get("vif", 0) + a
        return ((y - yp) ** 2).sum() + self.reg * (_l2(w["wi"]) + _l2(w["vif"]))
