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
    """P, Q and their Adam slots in HBM + the optimizer scalars (crec_bpr_tables / crec_adam)."""

    def __init__(self, sess, P0, Q0, lr, reg, adam_mode="tf1", beta1=0.9, beta2=0.999, eps=1e-8):
        self.sess = sess
        f32 = torch.float32
        self.P = sess.to_dev(P0, f32).contiguous()
        self.Q = sess.to_dev(Q0, f32).contiguous()
        self.mP, self.vP = torch.zeros_like(self.P), torch.zeros_like(self.P)
        self.mQ, self.vQ = torch.zeros_like(self.Q), torch.zeros_like(self.Q)
        self.tables = _lib.BprTables(*(t.data_ptr() for t in (self.P, self.Q, self.mP, self.vP, self.mQ, self.vQ)),
                                     self.P.shape[0], self.Q.shape[0], self.P.shape[1])
        mode = {"tf1": _lib.ADAM_TF1, "lazy": _lib.ADAM_LAZY}[adam_mode]
        self.adam = _lib.Adam(lr, beta1, beta2, eps, reg, mode, beta1, beta2, 0)

    def train_epoch(self, u, i, j, batch_size):
        """u, i, j: int32 device tensors [N].  Returns the per-batch summed losses (device fp32)."""
        s = self.sess
        n = int(u.numel())
        nb = -(-n // batch_size)
        loss = s.empty((nb,), torch.float32)
        s.ctx.check(s.lib.crec_bpr_train_epoch(s.ctx.h, C.byref(self.tables), C.byref(self.adam), s.ptr(u), s.ptr(i),
                                               s.ptr(j), n, int(batch_size), s.ptr(loss), s.stream_ptr()))
        return loss


def eval_topk(sess, P, Q, users, seen_ptr, seen_items, K, ascending=False):
    """crec_eval_topk -> (ids int32 [Bt,K], scores fp32 [Bt,K]) device tensors."""
    Bt = int(users.numel())
    ids = sess.empty((Bt, K), torch.int32)
    sc = sess.empty((Bt, K), torch.float32)
    sess.ctx.check(sess.lib.crec_eval_topk(sess.ctx.h, sess.ptr(P), sess.ptr(Q), Q.shape[0], Q.shape[1], sess.ptr(users),
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
