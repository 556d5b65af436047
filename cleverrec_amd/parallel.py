"""Multi-GPU form of the BPR path: one process per GPU of one node, RCCL over xGMI.

The path shards with one real exchange: users -- rows of P, their Adam slots, their
ui_train CSR, hence their sampler work and their triples -- are range-sharded over ranks, so all P
traffic is local; Q and its slots are replicated; every step each rank emits its dense partial dQ,
ONE all-reduce sums it, and every replica applies the identical tf1 Adam update
(crec_bpr_train_epoch_dp; replicas stay bit-identical).  torch.distributed is used only to ship
the RCCL unique id and to agree on the batch count."""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .data import RankingData
from .runtime import BprState


def user_range(user_nums, rank, world):
    """Contiguous user range [lo, hi) of a rank (balanced to within one user)."""
    base, rem = divmod(int(user_nums), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_users(data, rank, world):
    """This rank's slice of the `data` protocol: its users re-numbered from 0, items untouched."""
    lo, hi = user_range(data.user_nums, rank, world)

    def cut(rows):
        keys, ptr, items = rows
        sel = np.nonzero((keys >= lo) & (keys < hi))[0]
        if len(sel) == 0:
            return np.zeros(0, np.int32), np.zeros(1, np.int64), np.zeros(0, np.int32)
        a, b = sel[0], sel[-1] + 1  # keys ascending: a contiguous block
        return (keys[a:b] - lo).astype(np.int32), (ptr[a:b + 1] - ptr[a]).astype(np.int64), items[ptr[a]:ptr[b]].copy()
    return RankingData(hi - lo, data.item_nums, train_csr=cut(data.train_csr), test_csr_rows=cut(data._test_rows))


def broadcast_bytes(buf, src=0):
    """Ship a small byte string from `src` to every rank over the default process group (any backend)."""
    box = [bytes(buf) if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def agree_min(value):
    """min over ranks of a host integer (e.g. the number of full batches every rank can run)."""
    t = torch.tensor([int(value)], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


class DpBprState(BprState):
    """BprState whose train_epoch runs the data-parallel step.  P0: this rank's user rows; Q0: the
    replica (must be identical on every rank)."""

    def __init__(self, sess, P0, Q0, lr, reg, rank=None, world=None):
        super().__init__(sess, P0, Q0, lr, reg, "tf1")
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        uid = (C.c_char * 128)()
        if self.rank == 0:
            sess.ctx.check(sess.lib.crec_comm_unique_id(sess.ctx.h, uid))
        raw = bytes(uid) if self.world == 1 else broadcast_bytes(bytes(uid), 0)
        sess.ctx.check(sess.lib.crec_comm_init(sess.ctx.h, self.rank, self.world, C.c_char_p(raw)))

    def train_epoch(self, u, i, j, batch_size):
        """Collective.  u: row ids inside this rank's shard.  Returns per-batch losses summed over ranks."""
        s = self.sess
        n = int(u.numel())
        nb = -(-n // batch_size)
        loss = s.empty((nb,), torch.float32)
        s.ctx.check(s.lib.crec_bpr_train_epoch_dp(s.ctx.h, C.byref(self.tables), C.byref(self.adam), s.ptr(u), s.ptr(i),
                                                  s.ptr(j), n, int(batch_size), s.ptr(loss), s.stream_ptr()))
        return loss
