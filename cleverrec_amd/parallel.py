"""Multi-GPU forms of the BPR path: one process per GPU of one node, RCCL over xGMI.

Users -- rows of P, their Adam slots, their ui_train CSR, hence their sampler work and their triples -- are
range-sharded over ranks in both forms, so all P traffic is local.

  ShardedBprState  tables that outgrow a GPU (BASELINE configs[4]): items dealt round-robin (item g = local row g // G
                   of rank g % G); per step the referenced rows travel to the ranks that reference them and the
                   per-row gradient sums travel back to their owners, both as all-to-all-v over the xGMI mesh; the
                   owner applies lazy Adam (crec_bpr_train_epoch_sharded).  No dense all-reduce of a table.
  DpBprState       small catalogues with TF1-exact Adam (ml-1m shape): Q and its slots replicated, one all-reduce of the
                   dense dQ per step, identical tf1 update on every replica (crec_bpr_train_epoch_dp).

torch.distributed is used only to ship the RCCL unique id, to agree on the batch count and (evaluation) to gather
the item shards."""
import ctypes as C
import os

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

    def __init__(self, sess, P0, Q0, lr, reg, rank=None, world=None, transport="rccl", window_bytes=128 << 20):
        """transport: "rccl" (crec_comm_init), "peer" (the IPC windows of crec_comm_peer_*; window_bytes: an exchange carries up to half
        of it per rank), None (the ctx has a communicator)"""
        super().__init__(sess, P0, Q0, lr, reg, "tf1")
        if transport is None:
            self.rank, self.world = rank, world
        elif transport == "peer":
            self.rank, self.world = init_peer(sess, rank, world, window_bytes=window_bytes)
        else:
            self.rank, self.world = init_rccl(sess, rank, world)

    def train_epoch(self, u, i, j, batch_size, out=None):
        """Collective.  u: row ids inside this rank's shard.  Returns per-batch losses summed over ranks (in `out` if given)."""
        s = self.sess
        n = int(u.numel())
        nb = -(-n // batch_size)
        loss = s.empty((nb,), torch.float32) if out is None else out
        if loss.numel() < nb or loss.dtype != torch.float32 or not loss.is_contiguous():
            raise ValueError("out: a contiguous float32 device tensor of at least %d elements" % nb)
        s.ctx.check(s.lib.crec_bpr_train_epoch_dp(s.ctx.h, C.byref(self.tables), C.byref(self.adam), s.ptr(u), s.ptr(i),
                                                  s.ptr(j), n, int(batch_size), s.ptr(loss), s.stream_ptr()))
        return loss

    def synchronize(self):
        """Waits for the enqueued epochs and raises if an exchange gave up on a peer (crec_comm_status): train_epoch itself never
        drains the stream, it reports such a time-out one chunk or one call late."""
        s = self.sess
        s.ctx.check(s.lib.crec_comm_status(s.ctx.h, s.stream_ptr()))


def init_rccl(sess, rank=None, world=None):
    """crec_comm_init on this rank's ctx: rank 0 creates the RCCL unique id, torch.distributed ships it."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    uid = (C.c_char * 128)()
    if rank == 0:
        sess.ctx.check(sess.lib.crec_comm_unique_id(sess.ctx.h, uid))
    raw = bytes(uid) if world == 1 else broadcast_bytes(bytes(uid), 0)
    sess.ctx.check(sess.lib.crec_comm_init(sess.ctx.h, rank, world, C.c_char_p(raw)))
    return rank, world


PEER_BLOB_BYTES = 192


def init_peer(sess, rank=None, world=None, window_bytes=128 << 20):
    """The peer transport (include/crec.h "Peer transport"): this rank's window is created, torch.distributed ships the IPC
    handles, every rank maps every other rank's window.  An exchange carries up to window_bytes / 2 per rank."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    blob = (C.c_char * PEER_BLOB_BYTES)()
    sess.ctx.check(sess.lib.crec_comm_peer_create(sess.ctx.h, int(window_bytes), blob))
    if world == 1:
        blobs = [bytes(blob)]
    else:
        blobs = [None] * world
        dist.all_gather_object(blobs, bytes(blob))
    sess.ctx.check(sess.lib.crec_comm_peer_connect(sess.ctx.h, rank, world, C.c_char_p(b"".join(blobs))))
    if os.environ.get("CREC_PEER_FENCES") == "1":  # the textbook form of the window traffic (csrc/crec_peer.hip)
        sess.ctx.check(sess.lib.crec_comm_peer_fences(sess.ctx.h, 1))
    if world > 1:
        dist.barrier()  # nobody starts an exchange before everybody has mapped everybody
    sess.peer_window_bytes = int(window_bytes)
    return rank, world


def init_peer_local(sessions, window_bytes=128 << 20):
    """crec_comm_peer_connect_local: the peer transport among Sessions of THIS process (rank = position)."""
    arr = (C.c_void_p * len(sessions))(*[s.ctx.h for s in sessions])
    sessions[0].ctx.check(sessions[0].lib.crec_comm_peer_connect_local(arr, len(sessions), int(window_bytes)))


def init_local_group(sessions):
    """crec_comm_init_local: the ranks are the given Sessions of THIS process (rank = position), each to be driven by
    its own host thread."""
    arr = (C.c_void_p * len(sessions))(*[s.ctx.h for s in sessions])
    sessions[0].ctx.check(sessions[0].lib.crec_comm_init_local(arr, len(sessions)))


def shard_rows(item_nums, world):
    """rows per item shard: ceil(I / G)"""
    return -(-int(item_nums) // int(world))


def shard_items(Q, rank, world):
    """this rank's item shard of a full [I, d] table: rows rank, rank + G, ... padded with zeros to ceil(I / G) rows"""
    Q = np.asarray(Q)
    out = np.zeros((shard_rows(len(Q), world), Q.shape[1]), Q.dtype)
    part = Q[rank::world]
    out[:len(part)] = part
    return out


def unshard_items(shards, item_nums):
    """inverse of shard_items over all ranks' shards (numpy)"""
    world = len(shards)
    out = np.empty((int(item_nums), shards[0].shape[1]), shards[0].dtype)
    for r, sh in enumerate(shards):
        n = len(out[r::world])
        out[r::world] = np.asarray(sh)[:n]
    return out


def plan_batch_host(i, j, world, item_nums):
    """crec_shard_plan_host: one batch's index work on the host, with the arithmetic and the orders of the device kernels
    (shard_plan_batches + shard_pack_ids).  -> (req_keys [n_req], req_off [G+1], ci [nb], cj [nb], ids_send [n_req])"""
    lib = _lib.load()
    i, j = np.ascontiguousarray(i, np.int32), np.ascontiguousarray(j, np.int32)
    nb = len(i)
    keys, ids = np.empty(max(2 * nb, 1), np.int32), np.empty(max(2 * nb, 1), np.int32)
    off, ci, cj = np.empty(world + 1, np.int32), np.empty(max(nb, 1), np.int32), np.empty(max(nb, 1), np.int32)
    n = C.c_int32(0)
    rc = lib.crec_shard_plan_host(_lib._np(i), _lib._np(j), nb, int(world), int(item_nums), _lib._np(keys), C.byref(n), _lib._np(off),
                                  _lib._np(ci), _lib._np(cj), _lib._np(ids))
    if rc != _lib.OK:
        raise _lib.CrecError(rc, "crec_shard_plan_host")
    return keys[:n.value], off, ci[:nb], cj[:nb], ids[:n.value]


def owner_index_host(ids_recv, srv_cnt):
    """crec_shard_owner_index_host: the rows the peers ask of this owner in one batch (peer after peer) -> (order [n]: the
    slots sorted by row, peers in rank order within a row; urows [q]; useg [q+1]: where every row's slots start in order)"""
    lib = _lib.load()
    ids_recv, srv_cnt = np.ascontiguousarray(ids_recv, np.int32), np.ascontiguousarray(srv_cnt, np.int32)
    n = int(srv_cnt.sum())
    order, urows, useg = np.empty(max(n, 1), np.int32), np.empty(max(n, 1), np.int32), np.empty(n + 1, np.int32)
    q = C.c_int32(0)
    rc = lib.crec_shard_owner_index_host(_lib._np(ids_recv), _lib._np(srv_cnt), len(srv_cnt), _lib._np(order), _lib._np(urows),
                                         _lib._np(useg), C.byref(q))
    if rc != _lib.OK:
        raise _lib.CrecError(rc, "crec_shard_owner_index_host")
    return order[:n], urows[:q.value], useg[:q.value + 1]


class ShardedBprState(BprState):
    """P0: this rank's user rows; Q_shard: this rank's item shard (shard_items); item_nums: the global catalogue size.
    The communicator (init_rccl / init_local_group; none = a world of one) must exist before the first epoch."""

    def __init__(self, sess, P0, Q_shard, item_nums, lr, reg, rank=0, world=1):
        super().__init__(sess, P0, Q_shard, lr, reg, "lazy")
        self.item_nums, self.rank, self.world = int(item_nums), int(rank), int(world)
        assert self.Q.shape[0] >= shard_rows(item_nums, world)
        self.stats = _lib.ShardStats()

    def train_epoch(self, u, i, j, batch_size):
        """Collective.  u: row ids inside this rank's user shard; i, j: GLOBAL item ids (int32 device tensors).
        Returns the per-batch losses summed over ranks."""
        s = self.sess
        n = int(u.numel())
        loss = s.empty((-(-n // batch_size),), torch.float32)
        s.ctx.check(s.lib.crec_bpr_train_epoch_sharded(s.ctx.h, C.byref(self.tables), self.item_nums, C.byref(self.adam),
                                                       s.ptr(u), s.ptr(i), s.ptr(j), n, int(batch_size), s.ptr(loss),
                                                       C.byref(self.stats), s.stream_ptr()))
        return loss

    def stats_dict(self):
        return {k: int(getattr(self.stats, k)) for k, _ in _lib.ShardStats._fields_}


# ---- evaluation sharded over ranks: every rank ranks ITS test users against the full catalogue --------
def gather_item_shards(Q_shard, item_nums):
    """All-gather of the ranks' item shards into the full [I, d] table on this rank's device (one collective per
    evaluation, 0.5 GB at C5): rows come back in shard order and are dealt to their global positions r, r + G, ..."""
    world = dist.get_world_size()
    parts = [torch.empty_like(Q_shard) for _ in range(world)]
    dist.all_gather(parts, Q_shard.contiguous())
    out = torch.empty((int(item_nums), Q_shard.shape[1]), dtype=Q_shard.dtype, device=Q_shard.device)
    for r, sh in enumerate(parts):
        n = out[r::world].shape[0]
        out[r::world] = sh[:n]
    return out


def reduce_metric_sums(sums, count):
    """sums: float64 tensor [3, nk] of this rank's per-user HR / MRR / NDCG sums; count: its number of test users.
    -> (means [3, nk] over ALL ranks' users, total users).  One all-reduce of 3 * nk + 1 doubles."""
    buf = torch.cat([sums.reshape(-1).to(torch.float64), torch.tensor([float(count)], dtype=torch.float64, device=sums.device)])
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(buf)
    total = float(buf[-1].item())
    return (buf[:-1] / max(total, 1.0)).reshape(sums.shape), int(total)


def eval_rs_sharded(sess, P_local, Q_full, shard, topk, batch_size_t=1024, ascending=False):
    """RankingRecommender.test_model_rs (:198-247) with the test users sharded over ranks: this rank's users (rows of
    its P shard, its slice of the `data` protocol -- shard_users) against the full catalogue through crec_eval_topk /
    crec_rank_metrics, then one all-reduce of the metric sums.  -> (HR, MRR, NDCG) means per K over all ranks, users."""
    from . import runtime
    users = sess.to_dev(np.asarray(shard.test_users, dtype=np.int32))
    sp, si = sess.to_dev(shard.seen_csr[0]), sess.to_dev(shard.seen_csr[1] if len(shard.seen_csr[1]) else np.zeros(1, np.int32))
    tp, ti = sess.to_dev(shard.test_csr[0]), sess.to_dev(shard.test_csr[1] if len(shard.test_csr[1]) else np.zeros(1, np.int32))
    sums = torch.zeros((3, len(topk)), dtype=torch.float64, device=sess.device)
    for t0 in range(0, int(users.numel()), batch_size_t):
        cur = users[t0:t0 + batch_size_t]
        ids, _ = runtime.eval_topk(sess, P_local, Q_full, cur, sp, si, topk[-1], ascending)
        hr, mrr, ndcg = runtime.rank_metrics(sess, ids, cur, topk, tp, ti)
        sums += torch.stack([hr.sum(1), mrr.sum(1), ndcg.sum(1)])
    means, total = reduce_metric_sums(sums, int(users.numel()))
    return means[0].cpu().numpy(), means[1].cpu().numpy(), means[2].cpu().numpy(), total
