"""Session: what `sess` is in this build -- one GPU, one crec_ctx, torch as the allocator.

The reference opens a tf.Session on CUDA_VISIBLE_DEVICES=gpu.id (main.py:40-47) and hands it to
the model; here the same slot carries the HIP device binding.  torch is plumbing only: it owns
the HBM buffers and the stream; every computation goes through libcrec.so."""
import ctypes as C

import numpy as np
import torch

from . import _lib


class Session:
    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("cleverrec_amd needs an MI355X (gfx950): torch sees no GPU and there is no CPU path")
        self.device = torch.device("cuda", int(device))
        torch.cuda.set_device(self.device)
        self.ctx = _lib.Context(int(device))
        self.lib = self.ctx.lib

    # -- helpers -------------------------------------------------------------------------------
    def stream_ptr(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def to_dev(self, a, dtype=None):
        t = torch.as_tensor(np.ascontiguousarray(a))
        if dtype is not None:
            t = t.to(dtype)
        # pageable memory goes up at ~50 GB/s on this platform (33.6 MB of epoch sample: 0.64 ms); page-locking the sample
        # buffers was measured and dropped (tools/prof_prefetch.py: allocation jitter, a 137 ms first-touch registration)
        return t.to(self.device, non_blocking=False)

    def empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.device)

    @staticmethod
    def ptr(t):
        return C.c_void_p(t.data_ptr())

    def close(self):
        self.ctx.close()


class BprState:
    """P, Q and their optimizer slots in HBM + the optimizer scalars (crec_bpr_tables / crec_adam).
    adam_mode: 'tf1' | 'lazy' | 'deferred' (Adam; deferred = TF1 results with the lazy step's traffic, flush() before the
    tables are read), 'sgd', 'adagrad' (utils/tools.py:79-87)."""

    def __init__(self, sess, P0, Q0, lr, reg, adam_mode="tf1", beta1=0.9, beta2=0.999, eps=1e-8):
        self.sess = sess
        f32 = torch.float32
        P0, Q0 = np.asarray(P0, dtype=np.float32), np.asarray(Q0, dtype=np.float32)
        self.d = int(P0.shape[1])
        pad = (-self.d) % 4
        if pad:  # rows are walked as float4 slices: a width that is no multiple of 4 gets zero columns, which stay exactly zero
            # (their gradient -s (q_i - q_j) + reg p is 0, so m = v = 0 and Adam moves them by 0) and add 0 to every dot product
            P0, Q0 = np.pad(P0, ((0, 0), (0, pad))), np.pad(Q0, ((0, 0), (0, pad)))
        self.P = sess.to_dev(P0, f32).contiguous()
        self.Q = sess.to_dev(Q0, f32).contiguous()
        self.mP, self.vP = torch.zeros_like(self.P), torch.zeros_like(self.P)
        self.mQ, self.vQ = torch.zeros_like(self.Q), torch.zeros_like(self.Q)
        if adam_mode == "adagrad":  # tf.train.AdagradOptimizer: initial_accumulator_value = 0.1, kept in the `m` slots
            self.mP.fill_(0.1)
            self.mQ.fill_(0.1)
        self.tables = _lib.BprTables(*(t.data_ptr() for t in (self.P, self.Q, self.mP, self.vP, self.mQ, self.vQ)),
                                     self.P.shape[0], self.Q.shape[0], self.P.shape[1])
        mode = {"tf1": _lib.ADAM_TF1, "lazy": _lib.ADAM_LAZY, "sgd": _lib.OPT_SGD, "adagrad": _lib.OPT_ADAGRAD,
                "deferred": _lib.ADAM_DEFERRED}[adam_mode]
        self.adam = _lib.Adam(lr, beta1, beta2, eps, reg, mode, beta1, beta2, 0)
        self.deferred = None
        if adam_mode == "deferred":  # per row: the step its state is current at (crec_bpr_deferred)
            self.lastP = torch.zeros(self.P.shape[0], dtype=torch.int32, device=sess.device)
            self.lastQ = torch.zeros(self.Q.shape[0], dtype=torch.int32, device=sess.device)
            self.deferred = _lib.BprDeferred(self.lastP.data_ptr(), self.lastQ.data_ptr())

    def train_epoch(self, u, i, j, batch_size, out=None):
        """u, i, j: int32 device tensors [N].  Returns the per-batch summed losses (device fp32; written into `out` if given)."""
        s = self.sess
        n = int(u.numel())
        nb = -(-n // batch_size)
        loss = s.empty((nb,), torch.float32) if out is None else out
        if loss.numel() < nb or loss.dtype != torch.float32 or not loss.is_contiguous():
            raise ValueError("out: a contiguous float32 device tensor of at least %d elements" % nb)
        if getattr(self, 'deferred', None) is not None:
            s.ctx.check(s.lib.crec_bpr_train_epoch_deferred(s.ctx.h, C.byref(self.tables), C.byref(self.deferred), C.byref(self.adam),
                                                            s.ptr(u), s.ptr(i), s.ptr(j), n, int(batch_size), s.ptr(loss), s.stream_ptr()))
            return loss
        s.ctx.check(s.lib.crec_bpr_train_epoch(s.ctx.h, C.byref(self.tables), C.byref(self.adam), s.ptr(u), s.ptr(i),
                                               s.ptr(j), n, int(batch_size), s.ptr(loss), s.stream_ptr()))
        return loss

    def flush(self):
        """adam_mode 'deferred': replay the steps every row sat out (crec_bpr_adam_flush) -- before evaluating, saving or
        reading the tables.  A no-op in the other modes."""
        if getattr(self, 'deferred', None) is not None:
            s = self.sess
            s.ctx.check(s.lib.crec_bpr_adam_flush(s.ctx.h, C.byref(self.tables), C.byref(self.deferred), C.byref(self.adam), s.stream_ptr()))


def eval_topk(sess, P, Q, users, seen_ptr, seen_items, K, ascending=False, candidates=False):
    """crec_eval_topk (list = seen items, excluded) or crec_eval_topk_candidates (list = the only items that
    compete) -> (ids int32 [Bt,K], scores fp32 [Bt,K]) device tensors."""
    Bt = int(users.numel())
    ids = sess.empty((Bt, K), torch.int32)
    sc = sess.empty((Bt, K), torch.float32)
    fn = sess.lib.crec_eval_topk_candidates if candidates else sess.lib.crec_eval_topk
    sess.ctx.check(fn(sess.ctx.h, sess.ptr(P), sess.ptr(Q), Q.shape[0], Q.shape[1], sess.ptr(users),
                                           Bt, sess.ptr(seen_ptr), sess.ptr(seen_items), int(K), int(bool(ascending)),
                                           sess.ptr(ids), sess.ptr(sc), sess.stream_ptr()))
    return ids, sc


def rank_metrics(sess, topk_ids, users, ks, test_ptr, test_items):
    """crec_rank_metrics -> hr, mrr, ndcg float64 device tensors [len(ks), Bt]."""
    Bt, K = topk_ids.shape
    ks_h = np.ascontiguousarray(ks, dtype=np.int32)
    out = [sess.empty((len(ks_h), Bt), torch.float64) for _ in range(3)]
    sess.ctx.check(sess.lib.crec_rank_metrics(sess.ctx.h, sess.ptr(topk_ids), sess.ptr(users), Bt, K,
                                              ks_h.ctypes.data_as(C.c_void_p), len(ks_h), sess.ptr(test_ptr),
                                              sess.ptr(test_items), sess.ptr(out[0]), sess.ptr(out[1]), sess.ptr(out[2]),
                                              sess.stream_ptr()))
    torch.cuda.current_stream(sess.device).synchronize()  # ks_h is read by an async copy
    return out


def _optimizer_mode(optimizer):
    """configs['optimizer'] (utils/tools.py:79-87) -> crec_adam.adam_mode of the models that run on the shared row machinery."""
    try:
        # 'LazyAdam' (GMF / MLP / NeuMF, LRML; the model classes' adam_mode=lazy): Adam on the rows a batch touches only
        return {"adam": _lib.ADAM_TF1, "sgd": _lib.OPT_SGD, "adagrad": _lib.OPT_ADAGRAD, "lazyadam": _lib.ADAM_LAZY}[str(optimizer).lower()]
    except KeyError:
        raise ValueError("optimizer=%s: SGD | Adam | Adagrad" % optimizer)


def _zeros_slot(t, optimizer):
    """the m slot of a variable: Adam's first moment (0), or Adagrad's accumulator (initial_accumulator_value = 0.1)"""
    return torch.full_like(t, 0.1) if str(optimizer).lower() == "adagrad" else torch.zeros_like(t)


class NcfState:
    """Variables of GMF / MLP / NeuMF in HBM (crec_ncf): embedding tables + ONE flat block holding
    W_0|b_0|...|h with named views, Adam slots for all, and the optimizer scalars.

    kind: 'GMF' | 'MLP' | 'NeuMF'.  tables: dict with 'P_gmf','Q_gmf' and/or 'P_mlp','Q_mlp' (numpy/torch).
    Ws, bs: lists of tower weights/biases (MLP, NeuMF); h: head vector."""

    KINDS = {"GMF": _lib.NCF_GMF, "MLP": _lib.NCF_MLP, "NeuMF": _lib.NCF_NEUMF}

    LOSSES = {"cross_entropy": _lib.LOSS_CROSS_ENTROPY, "square": _lib.LOSS_SQUARE}

    def __init__(self, sess, kind, tables, Ws, bs, h, lr, reg_gmf=0.0, reg_mlp=0.0, beta1=0.9, beta2=0.999, eps=1e-8,
                 loss_func="cross_entropy", optimizer="Adam"):
        self.sess, self.kind = sess, kind
        f32 = torch.float32
        self.t, self.slots = {}, {}
        for k, a in tables.items():
            self.t[k] = sess.to_dev(a, f32).contiguous()
            self.slots[k] = (_zeros_slot(self.t[k], optimizer), torch.zeros_like(self.t[k]))
        flat = [np.asarray(x, dtype=np.float32).reshape(-1) for pair in zip(Ws, bs) for x in pair] + \
               [np.asarray(h, dtype=np.float32).reshape(-1)]
        self.small = sess.to_dev(np.concatenate(flat), f32)
        self.small_m, self.small_v = _zeros_slot(self.small, optimizer), torch.zeros_like(self.small)
        # named views into the flat block
        self.W, self.b, off = [], [], 0
        for W, b in zip(Ws, bs):
            W, b = np.asarray(W), np.asarray(b)
            self.W.append(self.small[off:off + W.size].view(*W.shape))
            off += W.size
            self.b.append(self.small[off:off + b.size])
            off += b.size
        self.h = self.small[off:]

        def param(name):
            if name not in self.t:
                return _lib.Param(None, None, None, 0, 0)
            t, (m, v) = self.t[name], self.slots[name]
            return _lib.Param(t.data_ptr(), m.data_ptr(), v.data_ptr(), t.shape[0], t.shape[1])
        small = _lib.Param(self.small.data_ptr(), self.small_m.data_ptr(), self.small_v.data_ptr(), 1, self.small.numel())
        self.model = _lib.Ncf(self.KINDS[kind], param("P_gmf"), param("Q_gmf"), param("P_mlp"), param("Q_mlp"), len(Ws), small,
                              reg_gmf, reg_mlp, self.LOSSES[loss_func])
        self.adam = _lib.Adam(lr, beta1, beta2, eps, 0.0, _optimizer_mode(optimizer), beta1, beta2, 0)

    def train_epoch(self, u, i, y, batch_size):
        """u, i int32, y float32 device tensors [N] -> per-batch summed losses (device fp32)."""
        s = self.sess
        n = int(u.numel())
        loss = s.empty((-(-n // batch_size),), torch.float32)
        s.ctx.check(s.lib.crec_ncf_train_epoch(s.ctx.h, C.byref(self.model), C.byref(self.adam), s.ptr(u), s.ptr(i), s.ptr(y),
                                               n, int(batch_size), s.ptr(loss), s.stream_ptr()))
        return loss


class FmState:
    """w0, wi, vif of model/rating/FM.py:27-31 with Adam slots in HBM (crec_fm)."""

    def __init__(self, sess, w0, wi, vif, lr, reg, beta1=0.9, beta2=0.999, eps=1e-8, optimizer="Adam"):
        self.sess = sess
        f32 = torch.float32
        self.w0 = sess.to_dev(np.asarray(w0, dtype=np.float32).reshape(1), f32)
        self.wi = sess.to_dev(np.asarray(wi, dtype=np.float32).reshape(-1), f32)
        self.vif = sess.to_dev(vif, f32).contiguous()
        self.slots = {k: (_zeros_slot(t, optimizer), torch.zeros_like(t)) for k, t in (("w0", self.w0), ("wi", self.wi), ("vif", self.vif))}

        def param(name, t, rows, cols):
            m, v = self.slots[name]
            return _lib.Param(t.data_ptr(), m.data_ptr(), v.data_ptr(), rows, cols)
        self.model = _lib.Fm(param("w0", self.w0, 1, 1), param("wi", self.wi, self.wi.numel(), 1),
                             param("vif", self.vif, self.vif.shape[0], self.vif.shape[1]), reg)
        self.adam = _lib.Adam(lr, beta1, beta2, eps, 0.0, _optimizer_mode(optimizer), beta1, beta2, 0)

    def train_epoch(self, x_idx, x_val, y, batch_size):
        """x_idx int32 [N,F], x_val fp32 [N,F], y fp32 [N] (device) -> (pre-update y_pre [N], per-batch losses)."""
        s = self.sess
        n, F = x_idx.shape
        y_pre = s.empty((n,), torch.float32)
        loss = s.empty((-(-n // batch_size),), torch.float32)
        s.ctx.check(s.lib.crec_fm_train_epoch(s.ctx.h, C.byref(self.model), C.byref(self.adam), s.ptr(x_idx), s.ptr(x_val),
                                              s.ptr(y), n, F, int(batch_size), s.ptr(y_pre), s.ptr(loss), s.stream_ptr()))
        return y_pre, loss

    def predict(self, x_idx, x_val):
        s = self.sess
        n, F = x_idx.shape
        y_pre = s.empty((n,), torch.float32)
        s.ctx.check(s.lib.crec_fm_predict(s.ctx.h, C.byref(self.model), s.ptr(x_idx), s.ptr(x_val), n, F, s.ptr(y_pre),
                                          s.stream_ptr()))
        return y_pre


def rmse_mae(sess, y, y_pre):
    """cal_rmse_mae (utils/metrics.py:22-29) on device tensors, float64 -> (rmse, mae) host floats."""
    out = sess.empty((2,), torch.float64)
    sess.ctx.check(sess.lib.crec_rmse_mae(sess.ctx.h, sess.ptr(y), sess.ptr(y_pre), int(y.numel()), sess.ptr(out),
                                          sess.stream_ptr()))
    r = out.cpu().numpy()
    return float(r[0]), float(r[1])


class CmlState:
    """P, Q of model/ranking/CML.py:28-32 with Adam slots in HBM (crec_cml)."""

    def __init__(self, sess, P0, Q0, lr, reg, margin, beta1=0.9, beta2=0.999, eps=1e-8, optimizer="Adam"):
        self.sess = sess
        f32 = torch.float32
        self.P, self.Q = sess.to_dev(P0, f32).contiguous(), sess.to_dev(Q0, f32).contiguous()
        self.mP, self.vP = _zeros_slot(self.P, optimizer), torch.zeros_like(self.P)
        self.mQ, self.vQ = _zeros_slot(self.Q, optimizer), torch.zeros_like(self.Q)
        par = lambda t, m, v: _lib.Param(t.data_ptr(), m.data_ptr(), v.data_ptr(), t.shape[0], t.shape[1])
        self.model = _lib.Cml(par(self.P, self.mP, self.vP), par(self.Q, self.mQ, self.vQ), reg, margin)
        self.adam = _lib.Adam(lr, beta1, beta2, eps, 0.0, _optimizer_mode(optimizer), beta1, beta2, 0)

    def train_epoch(self, u, i, neg, batch_size):
        """u, i int32 [N], neg int32 [N, K] (device) -> per-batch losses (device fp32)."""
        s = self.sess
        n, K = neg.shape
        loss = s.empty((-(-n // batch_size),), torch.float32)
        s.ctx.check(s.lib.crec_cml_train_epoch(s.ctx.h, C.byref(self.model), C.byref(self.adam), s.ptr(u), s.ptr(i), s.ptr(neg),
                                               n, K, int(batch_size), s.ptr(loss), s.stream_ptr()))
        return loss

    def eval_tables(self, clip_users=True):
        """(P_aug, Q_aug) for crec_eval_topk(ascending=1): CML._predict's distances as a dot product.  clip_users: the rs
        branch scores with clip_1(p_u) (CML.py:76,84), the loo / sampled-negatives branch with p_u itself (CML.py:42,81-82)."""
        s = self.sess
        U, d = self.P.shape
        I = self.Q.shape[0]
        Pa, Qa = s.empty((U, d + 4), torch.float32), s.empty((I, d + 4), torch.float32)
        s.ctx.check(s.lib.crec_cml_eval_tables(s.ctx.h, s.ptr(self.P), s.ptr(self.Q), U, I, d, int(bool(clip_users)), s.ptr(Pa),
                                               s.ptr(Qa), s.stream_ptr()))
        return Pa, Qa


class LrmlState:
    """P, Q, K, M of model/ranking/LRML.py:24-34 with Adam slots in HBM (crec_lrml); K and M are views into ONE flat
    block [K | M]."""

    def __init__(self, sess, P0, Q0, K0, M0, lr, reg, margin, beta1=0.9, beta2=0.999, eps=1e-8, optimizer="Adam"):
        self.sess = sess
        f32 = torch.float32
        self.P, self.Q = sess.to_dev(P0, f32).contiguous(), sess.to_dev(Q0, f32).contiguous()
        K0, M0 = np.asarray(K0, dtype=np.float32), np.asarray(M0, dtype=np.float32)
        d, mem = K0.shape
        if M0.shape != (mem, d) or self.P.shape[1] != d or self.Q.shape[1] != d:
            raise ValueError("LRML: K [embed_size, mem_size], M [mem_size, embed_size]")
        self.small = sess.to_dev(np.concatenate([K0.reshape(-1), M0.reshape(-1)]), f32)
        self.K, self.M = self.small[:d * mem].view(d, mem), self.small[d * mem:].view(mem, d)
        self.slots = {k: (_zeros_slot(t, optimizer), torch.zeros_like(t)) for k, t in (("P", self.P), ("Q", self.Q), ("small", self.small))}

        def par(name, t, rows, cols):
            m, v = self.slots[name]
            return _lib.Param(t.data_ptr(), m.data_ptr(), v.data_ptr(), rows, cols)
        self.model = _lib.Lrml(par("P", self.P, self.P.shape[0], d), par("Q", self.Q, self.Q.shape[0], d),
                               par("small", self.small, 1, self.small.numel()), mem, reg, margin)
        self.adam = _lib.Adam(lr, beta1, beta2, eps, 0.0, _optimizer_mode(optimizer), beta1, beta2, 0)

    def train_epoch(self, u, i, j, batch_size):
        """u, i, j int32 [N] (device) -> per-batch summed losses (device fp32)."""
        s = self.sess
        n = int(u.numel())
        loss = s.empty((-(-n // batch_size),), torch.float32)
        s.ctx.check(s.lib.crec_lrml_train_epoch(s.ctx.h, C.byref(self.model), C.byref(self.adam), s.ptr(u), s.ptr(i), s.ptr(j),
                                                n, int(batch_size), s.ptr(loss), s.stream_ptr()))
        return loss

    def scores(self, users):
        """users int32 [Bt] (device) -> [Bt, I] squared distances (LRML.py:74-76); rank ascending."""
        s = self.sess
        Bt = int(users.numel())
        out = s.empty((Bt, self.Q.shape[0]), torch.float32)
        s.ctx.check(s.lib.crec_lrml_scores(s.ctx.h, C.byref(self.model), s.ptr(users), Bt, s.ptr(out), s.stream_ptr()))
        return out


class TranscfState:
    """P, Q of model/ranking/TransCF.py:27-32 with Adam slots and the two neighbourhood CSRs (get_sp_mat) in HBM
    (crec_transcf)."""

    def __init__(self, sess, P0, Q0, sp_mat, lr, reg1, reg2, margin, beta1=0.9, beta2=0.999, eps=1e-8, optimizer="Adam"):
        self.sess = sess
        f32 = torch.float32
        self.P, self.Q = sess.to_dev(P0, f32).contiguous(), sess.to_dev(Q0, f32).contiguous()
        if str(optimizer).lower() == "lazyadam":
            raise ValueError("TransCF: both tables take dense gradients (the neighbourhood products reach every row): no lazy Adam")
        self.slots = {k: (_zeros_slot(t, optimizer), torch.zeros_like(t)) for k, t in (("P", self.P), ("Q", self.Q))}
        ui_ptr, ui_items, iu_ptr, iu_users = sp_mat
        U, I = self.P.shape[0], self.Q.shape[0]
        if len(ui_ptr) != U + 1 or len(iu_ptr) != I + 1 or ui_ptr[-1] != iu_ptr[-1]:
            raise ValueError("TransCF: ui_ptr [U+1], iu_ptr [I+1] over the same interactions")
        pad = lambda a: a if len(a) else np.zeros(1, a.dtype)
        self.csr = tuple(sess.to_dev(np.ascontiguousarray(a)) for a in
                         (np.asarray(ui_ptr, np.int64), pad(np.asarray(ui_items, np.int32)), np.asarray(iu_ptr, np.int64),
                          pad(np.asarray(iu_users, np.int32))))

        def par(name, t):
            m, v = self.slots[name]
            return _lib.Param(t.data_ptr(), m.data_ptr(), v.data_ptr(), t.shape[0], t.shape[1])
        self.model = _lib.Transcf(par("P", self.P), par("Q", self.Q), *(c.data_ptr() for c in self.csr), reg1, reg2, margin)
        self.adam = _lib.Adam(lr, beta1, beta2, eps, 0.0, _optimizer_mode(optimizer), beta1, beta2, 0)
        self._nbr_ws = None

    def train_epoch(self, u, i, j, batch_size):
        """u, i, j int32 [N] (device) -> per-batch summed losses (device fp32)."""
        s = self.sess
        n = int(u.numel())
        loss = s.empty((-(-n // batch_size),), torch.float32)
        s.ctx.check(s.lib.crec_transcf_train_epoch(s.ctx.h, C.byref(self.model), C.byref(self.adam), s.ptr(u), s.ptr(i), s.ptr(j),
                                                   n, int(batch_size), s.ptr(loss), s.stream_ptr()))
        return loss

    def scores(self, users, clip_users=True):
        """users int32 [Bt] (device) -> [Bt, I] squared distances (TransCF.py:85-91); rank ascending."""
        s = self.sess
        Bt = int(users.numel())
        U, d = self.P.shape
        I = self.Q.shape[0]
        if self._nbr_ws is None:
            self._nbr_ws = s.empty(((U + I) * d,), torch.float32)
        out = s.empty((Bt, I), torch.float32)
        s.ctx.check(s.lib.crec_transcf_scores(s.ctx.h, C.byref(self.model), s.ptr(users), Bt, int(bool(clip_users)),
                                              s.ptr(self._nbr_ws), s.ptr(out), s.stream_ptr()))
        return out


class SbprState:
    """P, Q, bias of model/ranking/SBPR.py:31-37 with Adam slots in HBM (crec_sbpr)."""

    def __init__(self, sess, P0, Q0, bias0, lr, reg, beta1=0.9, beta2=0.999, eps=1e-8, optimizer="Adam"):
        self.sess = sess
        f32 = torch.float32
        self.P, self.Q = sess.to_dev(P0, f32).contiguous(), sess.to_dev(Q0, f32).contiguous()
        self.bias = sess.to_dev(np.asarray(bias0, dtype=np.float32).reshape(-1), f32)
        self.slots = {k: (_zeros_slot(t, optimizer), torch.zeros_like(t)) for k, t in (("P", self.P), ("Q", self.Q), ("bias", self.bias))}

        def par(name, t, rows, cols):
            m, v = self.slots[name]
            return _lib.Param(t.data_ptr(), m.data_ptr(), v.data_ptr(), rows, cols)
        self.model = _lib.Sbpr(par("P", self.P, self.P.shape[0], self.P.shape[1]), par("Q", self.Q, self.Q.shape[0], self.Q.shape[1]),
                               par("bias", self.bias, self.bias.numel(), 1), reg)
        self.adam = _lib.Adam(lr, beta1, beta2, eps, 0.0, _optimizer_mode(optimizer), beta1, beta2, 0)

    def train_epoch(self, u, i, k, j, suk, batch_size):
        """u, i, k, j int32 [N], suk fp32 [N] (device) -> per-batch losses (device fp32)."""
        s = self.sess
        n = int(u.numel())
        loss = s.empty((-(-n // batch_size),), torch.float32)
        s.ctx.check(s.lib.crec_sbpr_train_epoch(s.ctx.h, C.byref(self.model), C.byref(self.adam), s.ptr(u), s.ptr(i), s.ptr(k),
                                                s.ptr(j), s.ptr(suk), n, int(batch_size), s.ptr(loss), s.stream_ptr()))
        return loss


class CuneState(SbprState):
    """CUNE_BPR's variables (model/ranking/CUNE_BPR.py:33-42): SBPR's P, Q, bias plus the trainable scalar s (crec_cune)."""

    def __init__(self, sess, P0, Q0, bias0, s0, lr, reg, beta1=0.9, beta2=0.999, eps=1e-8, optimizer="Adam"):
        super().__init__(sess, P0, Q0, bias0, lr, reg, beta1, beta2, eps, optimizer)
        self.s = sess.to_dev(np.asarray(s0, dtype=np.float32).reshape(1), torch.float32)
        self.slots["s"] = (_zeros_slot(self.s, optimizer), torch.zeros_like(self.s))
        sb = self.model
        self.model = _lib.Cune(sb.P, sb.Q, sb.bias, _lib.Param(self.s.data_ptr(), self.slots["s"][0].data_ptr(), self.slots["s"][1].data_ptr(), 1, 1), reg)

    def train_epoch(self, u, i, k, j, batch_size):
        """u, i, k, j int32 [N] (device) -> per-batch losses (device fp32)."""
        s = self.sess
        n = int(u.numel())
        loss = s.empty((-(-n // batch_size),), torch.float32)
        s.ctx.check(s.lib.crec_cune_train_epoch(s.ctx.h, C.byref(self.model), C.byref(self.adam), s.ptr(u), s.ptr(i), s.ptr(k), s.ptr(j),
                                                n, int(batch_size), s.ptr(loss), s.stream_ptr()))
        return loss


def ncf_scores(sess, state, users):
    """crec_ncf_scores -> logits [Bt, I] (device fp32); pre_scores = sigmoid(logits) ranks identically."""
    Bt = int(users.numel())
    I = (state.t["Q_mlp"] if "Q_mlp" in state.t else state.t["Q_gmf"]).shape[0]
    out = sess.empty((Bt, I), torch.float32)
    sess.ctx.check(sess.lib.crec_ncf_scores(sess.ctx.h, C.byref(state.model), sess.ptr(users), Bt, sess.ptr(out), sess.stream_ptr()))
    return out


TOPK_SCORES_MAX_ITEMS = 40000  # crec_topk_scores stages a user's score row in LDS
_chunk_lists = {}


def _topk_scores_call(sess, scores, users, seen_ptr, seen_items, K, ascending, candidates):
    Bt, I = scores.shape
    ids, sc = sess.empty((Bt, K), torch.int32), sess.empty((Bt, K), torch.float32)
    fn = sess.lib.crec_topk_scores_candidates if candidates else sess.lib.crec_topk_scores
    sess.ctx.check(fn(sess.ctx.h, sess.ptr(scores), I, sess.ptr(users), Bt, sess.ptr(seen_ptr),
                      sess.ptr(seen_items), int(K), int(bool(ascending)), sess.ptr(ids), sess.ptr(sc),
                      sess.stream_ptr()))
    return ids, sc


def topk_scores(sess, scores, users, seen_ptr, seen_items, K, ascending=False, candidates=False):
    """crec_topk_scores[_candidates] over a materialised [Bt, I] score matrix -> (ids int32 [Bt,K], scores fp32 [Bt,K]).
    Catalogues beyond 40 000 items (one score row no longer fits the kernel's LDS stage) are ranked a stretch of the item
    axis at a time -- the users' lists cut to the stretch -- and the stretches' K best are ranked once more; ties keep going to
    the lower item id (lower stretch first, and inside a stretch the kernel's own order)."""
    Bt, I = scores.shape
    if I <= TOPK_SCORES_MAX_ITEMS:
        return _topk_scores_call(sess, scores, users, seen_ptr, seen_items, K, ascending, candidates)
    # the cache holds the tensors themselves and compares identity: addresses alone come back (the caching allocator hands a freed
    # block to the next dataset of a grid search), and a stale cut would filter the wrong items
    held = _chunk_lists.get("lists")
    if held is None or held[0] is not seen_ptr or held[1] is not seen_items or held[2] != I:  # the lists per stretch, on the host once (CSR by user id, items ascending)
        ptr, items = seen_ptr.cpu().numpy(), seen_items.cpu().numpy()
        row = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
        cuts = []
        for lo in range(0, I, TOPK_SCORES_MAX_ITEMS):
            hi = min(lo + TOPK_SCORES_MAX_ITEMS, I)
            m = (items[:len(row)] >= lo) & (items[:len(row)] < hi)
            p = np.zeros(len(ptr), np.int64)
            np.cumsum(np.bincount(row[m], minlength=len(ptr) - 1), out=p[1:])
            it = (items[:len(row)][m] - lo).astype(np.int32)
            cuts.append((lo, hi, sess.to_dev(p), sess.to_dev(it if len(it) else np.zeros(1, np.int32))))
        _chunk_lists.clear()
        _chunk_lists["lists"] = (seen_ptr, seen_items, I, cuts)
    parts_i, parts_s = [], []
    for lo, hi, p, it in _chunk_lists["lists"][3]:
        ids, sc = _topk_scores_call(sess, scores[:, lo:hi].contiguous(), users, p, it, K, ascending, candidates)
        parts_i.append(torch.where(ids >= 0, ids + lo, ids))
        parts_s.append(sc)
    cand_ids, cand_sc = torch.cat(parts_i, 1), torch.cat(parts_s, 1).contiguous()  # [Bt, stretches * K], lower ids first among equals
    none = sess.zeros((len(seen_ptr),), torch.int64)  # nobody has a list in this round (sized by the CSR: no read-back of users.max())
    cols, sc = _topk_scores_call(sess, cand_sc, users, none, sess.zeros((1,), torch.int32), K, ascending, False)
    ids = torch.gather(cand_ids, 1, cols.clamp(min=0).long()).to(torch.int32)
    return torch.where(cols >= 0, ids, cols), sc


class DeviceSampler:
    """Epoch samplers on the GPU: the acceptance rule of utils/sampler.py (uniform negatives, not among the user's
    training items, no repeat inside a positive's group), a counter-based generator instead of NumPy's MT19937
    stream -- statistically equivalent, NOT stream-compatible (the bit-exact host samplers stay the default).
    One launch draws every negative of the epoch (crec_sample_negatives_device); the epoch shuffle is a device
    randperm and the three output formats are gathers.  Opt in with `sampler_mode=device` in the .properties file."""

    def __init__(self, sess, data, seed=0):
        from .data import RankingData
        if not hasattr(data, 'train_csr'):
            data = RankingData(data.user_nums, data.item_nums, data.ui_train, data.ui_test)
        self.sess, self.item_nums, self.seed, self.epoch = sess, int(data.item_nums), int(seed), 0
        keys, ptr, items = data.train_csr
        self.n_pos = int(ptr[-1])
        self.edge_user = sess.to_dev(np.repeat(np.asarray(keys, np.int32), np.diff(ptr)).astype(np.int32))
        self.items = sess.to_dev(np.asarray(items, np.int32))
        sp, si = data.seen_csr
        self.seen_ptr = sess.to_dev(sp)
        self.seen_items = sess.to_dev(si if len(si) else np.zeros(1, np.int32))
        self.max_seen = C.c_int32(int(np.diff(sp).max()) if len(sp) > 1 else 0)
        self.gen = torch.Generator(device=sess.device)
        self.gen.manual_seed(self.seed)

    def negatives(self, neg_ratio):
        """[n_pos, neg_ratio] int32: the negatives of every training positive, in ui_train order."""
        s = self.sess
        neg = s.empty((max(self.n_pos, 1), neg_ratio), torch.int32)
        s.ctx.check(s.lib.crec_sample_negatives_device(s.ctx.h, self.seed, self.epoch, self.n_pos, s.ptr(self.edge_user),
                                                       s.ptr(self.seen_ptr), s.ptr(self.seen_items), C.byref(self.max_seen),
                                                       self.item_nums, int(neg_ratio), s.ptr(neg), s.stream_ptr()))
        self.epoch += 1
        return neg[:self.n_pos]

    def _perm(self, n):
        return torch.randperm(n, device=self.sess.device, generator=self.gen)

    def pairwise(self, neg_ratio):
        """(u, i, j) int32 [n_pos * neg_ratio], shuffled (pairwise_ranking_sampler, sampler.py:46-74)."""
        neg = self.negatives(neg_ratio)
        perm = self._perm(self.n_pos * neg_ratio)
        e = torch.div(perm, neg_ratio, rounding_mode='floor')
        return self.edge_user[e].contiguous(), self.items[e].contiguous(), neg.reshape(-1)[perm].contiguous()

    def pointwise(self, neg_ratio):
        """(u, i, y): every positive with label 1 and its negatives with label 0, shuffled (sampler.py:10-43)."""
        neg = self.negatives(neg_ratio)
        w = 1 + neg_ratio
        perm = self._perm(self.n_pos * w)
        e, t = torch.div(perm, w, rounding_mode='floor'), perm % w
        both = torch.cat([self.items[:, None], neg], dim=1)  # column 0: the positive, 1..: its negatives
        return self.edge_user[e].contiguous(), both.reshape(-1)[perm].contiguous(), (t == 0).to(torch.float32)

    def cml(self, neg_ratio):
        """(u, i, neg[n_pos, neg_ratio]), shuffled (ranking_sampler_cml, sampler.py:77-99)."""
        neg = self.negatives(neg_ratio)
        perm = self._perm(self.n_pos)
        return self.edge_user[perm].contiguous(), self.items[perm].contiguous(), neg[perm].contiguous()


class MtSampler:
    """The reference's epoch samplers ON the device, bit for bit (crec_dsampler_*, csrc/mt_sampler.hip): NumPy's MT19937
    word stream produced by jump-ahead chains, randint's rejection as a compaction, the per-user redraw loop and the
    shuffle's draws as tile walks, the permutation as a sort + pointer chase.  Same columns, same np.random state
    afterwards as utils/sampler.py:10-99 -- and the columns never leave HBM.

    next_epoch(kind, neg_ratio) is the drop-in: it reads np.random's state, returns the sample the reference would
    draw now and leaves np.random where the reference would.  With ahead=True the sample after this one is enqueued
    right away (from the state this one ends in, on the caller's stream: it runs in front of this epoch's training steps
    and the host never waits for it); it is used at the next call only if np.random still stands exactly there."""

    KINDS = {"pairwise": 0, "pointwise": 1, "cml": 2}

    def __init__(self, sess, data):
        from .data import RankingData
        if not hasattr(data, "train_csr"):
            data = RankingData(data.user_nums, data.item_nums, data.ui_train, data.ui_test)
        keys, ptr, items = data.train_csr
        keys, ptr = np.ascontiguousarray(keys, np.int32), np.ascontiguousarray(ptr, np.int64)
        items = np.ascontiguousarray(items if len(items) else np.zeros(1, np.int32), np.int32)
        self.sess, self.n_pos, self.item_nums = sess, int(ptr[-1]), int(data.item_nums)
        self.h = C.c_void_p()
        sess.ctx.check(sess.lib.crec_dsampler_create(sess.ctx.h, _lib._np(ptr), _lib._np(items), len(keys), _lib._np(keys),
                                                     self.item_nums, C.byref(self.h), sess.stream_ptr()))
        self.side = torch.cuda.Stream(device=sess.device)
        self._flight = None  # (kind, neg_ratio, key, pos, (u, i, third), stream) of the sample enqueued and not yet collected

    def close(self):
        if getattr(self, "h", None):
            self.sess.lib.crec_dsampler_destroy(self.h)
            self.h = None

    __del__ = close

    def _shapes(self, kind, nr):
        n = {"pairwise": self.n_pos * nr, "pointwise": self.n_pos * (1 + nr), "cml": self.n_pos}[kind]
        third = (torch.float32, (n,)) if kind == "pointwise" else (torch.int32, (n, nr) if kind == "cml" else (n,))
        return n, third

    def enqueue(self, kind, neg_ratio, key, pos, stream=None):
        """start one sample from the generator state (key, pos) on `stream` (default: the current one)"""
        if self._flight is not None:
            raise RuntimeError("MtSampler: collect() the sample in flight first")
        s = self.sess
        stream = stream or torch.cuda.current_stream(s.device)
        n, (tdt, tshape) = self._shapes(kind, int(neg_ratio))
        with torch.cuda.stream(stream):
            u, i = (torch.empty(max(n, 1), dtype=torch.int32, device=s.device) for _ in range(2))
            third = torch.empty([max(x, 1) for x in tshape], dtype=tdt, device=s.device)
            key = np.ascontiguousarray(key, dtype=np.uint32)
            s.ctx.check(s.lib.crec_dsampler_sample(s.ctx.h, self.h, self.KINDS[kind], int(neg_ratio), _lib._np(key), int(pos),
                                                   s.ptr(u), s.ptr(i), s.ptr(third), C.c_void_p(stream.cuda_stream)))
        self._flight = (kind, int(neg_ratio), key.copy(), int(pos), (u, i, third), stream, n, tshape)

    def collect(self):
        """-> ((u, i, third) device tensors, (key, pos) generator state after the sample); waits for the sample"""
        kind, nr, _, _, (u, i, third), stream, n, tshape = self._flight
        s = self.sess
        n_out, key, pos = C.c_int64(), np.empty(624, dtype=np.uint32), C.c_int32()
        s.ctx.check(s.lib.crec_dsampler_finish(s.ctx.h, self.h, C.byref(n_out), _lib._np(key), C.byref(pos)))
        self._flight = None
        assert n_out.value == n
        cur = torch.cuda.current_stream(s.device)
        if stream != cur:  # the sample is complete (finish waited for it); tell the allocator who reads the columns now
            for t in (u, i, third):
                t.record_stream(cur)
        third = third[:n] if kind != "cml" else third[:n, :nr]
        return (u[:n], i[:n], third), (key, pos.value)

    def stats(self):
        out = (C.c_int64 * 13)()
        self.sess.ctx.check(self.sess.lib.crec_dsampler_stats(self.h, out))
        return dict(words_negatives=out[0], words_shuffle=out[1], candidates=out[2], candidates_used=out[3],
                    neg_tiles=out[4], neg_clocks=out[5], neg_us=out[6] / 100.0, perm_tiles=out[7], perm_clocks=out[8], perm_us=out[9] / 100.0, walk_fallbacks=out[10], last_fallback=(out[11], out[12]))

    def sample(self, kind, neg_ratio, key, pos):
        """one sample from an explicit generator state, synchronously"""
        self.enqueue(kind, neg_ratio, key, pos)
        return self.collect()

    def drop(self):
        if self._flight is not None:
            self.collect()

    def next_epoch(self, kind, neg_ratio, ahead=True):
        st = np.random.get_state()
        assert st[0] == "MT19937"
        cols = None
        fl = self._flight
        if fl is not None:
            hit = fl[0] == kind and fl[1] == int(neg_ratio) and fl[3] == st[2] and np.array_equal(fl[2], st[1])
            got = self.collect()
            if hit:
                cols, end = got
        if cols is None:
            cols, end = self.sample(kind, neg_ratio, st[1], st[2])
        np.random.set_state(("MT19937", end[0], end[1], st[3], st[4]))  # no normals drawn: the cached gaussian stays
        if ahead:
            # on the CALLER's stream, in front of this epoch's steps -- not beside them.  The host still never waits for a sample, and
            # the GPU time is the same (an epoch's steps take 2.3 ms since they run as dataflow launches, a sample 2.2 ms: one after the
            # other 4.5 ms, beside each other they took 4.8); but a bpr_steps_flow launch needs every one of its workgroups on the chip
            # at once, and beside another stream's kernels with 1024-thread workgroups -- the sampler's walks and sorts -- the two
            # launches can keep each other's last workgroups off the chip for ever (seen: the launch gave up after its time-out)
            self.enqueue(kind, neg_ratio, end[0], end[1], stream=None)
        return cols
