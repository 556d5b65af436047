"""CPU restatement of the row-sharded exchange of include/crec.h ("row-sharded training") over torch.distributed
-- TEST INFRASTRUCTURE ONLY (tests/test_parallel_cpu.py runs it on gloo, world sizes 2 and 4).

Each rank holds its user rows and its item shard only (item g = local row g // G of rank g % G).  Per step:
  plan     the batch's item ids -> owner-major keys (g % G) * S + g // G; the sorted unique keys are the request
           list (grouped by owner), every i / j becomes its rank in that list (the compact id)      [shard_plan_batches]
  ids      all-to-all-v of the request lists (local row ids) to the owners                           [shard_pack_ids]
  forward  every owner sends the requested rows, peer by peer, in request order (all-to-all-v)       [shard_serve_rows]
  step     the rank's lazy BPR step on its own triples against the compact table: Adam on its user rows, per compact
           row the gradient sum incl. the l2 term (C oracle, orc_bpr_shard_local)
  back     the gradient rows return to their owners (the forward sizes, reversed)
  apply    the owner adds a row's gradient rows in peer order and applies lazy Adam                   [shard_owner_apply]
The HIP path exchanges the request lists once per chunk of batches; the result is the same."""
import numpy as np
import torch
import torch.distributed as dist

from . import oracle as O


def plan(i, j, G, S):
    """-> (keys: sorted unique owner-major keys, off [G+1]: first rank of every owner's segment, ci, cj)"""
    g = np.concatenate([i, j]).astype(np.int64)
    keys, inv = np.unique((g % G) * S + g // G, return_inverse=True)
    off = np.searchsorted(keys, np.arange(G + 1) * S).astype(np.int64)
    return keys, off, inv[:len(i)].astype(np.int32), inv[len(i):].astype(np.int32)


def _a2a(send, scnt, rcnt, width, dtype):
    out = torch.empty((int(sum(rcnt)), width) if width else (int(sum(rcnt)),), dtype=dtype)
    dist.all_to_all_single(out, torch.as_tensor(send), output_split_sizes=[int(c) for c in rcnt],
                           input_split_sizes=[int(c) for c in scnt])
    return out.numpy()


class ShardedRank:
    """One rank's state: P_local [U_loc, d], Q_shard [S, d] with Adam slots; lr schedule shared by construction."""

    def __init__(self, P_local, Q_shard, item_nums, lr=1e-3, reg=0.01, planner=None):
        self.G, self.rank = dist.get_world_size(), dist.get_rank()
        self.S = -(-int(item_nums) // self.G)
        self.item_nums, self.planner = int(item_nums), planner
        self.user = O.BprModel(P_local, np.zeros((1, P_local.shape[1]), np.float32), lr=lr, reg=reg, adam_mode=1)
        self.item = O.BprModel(np.zeros((1, P_local.shape[1]), np.float32), Q_shard, lr=lr, reg=reg, adam_mode=1)

    def step(self, u, i, j):
        G, S, d = self.G, self.S, self.user.P.shape[1]
        if self.planner is None:
            keys, off, ci, cj = plan(i, j, G, S)
            local = (keys - (keys // S) * S).astype(np.int64)
        else:
            keys, off, ci, cj, local = self.planner.plan(i, j, G, self.item_nums)
            off, local = off.astype(np.int64), local.astype(np.int64)
        cnt = np.diff(off)
        srv_cnt = _a2a(cnt, [1] * G, [1] * G, 0, torch.int64)                      # how many rows every peer asks of me
        srv_rows = _a2a(local, cnt, srv_cnt, 0, torch.int64)                       # which rows, peer-major
        Qc = _a2a(self.item.Q[srv_rows], srv_cnt, cnt, d, torch.float32)           # forward: my compact table
        lr_t = float(self.item.sched.next(1)[0])
        m = self.user
        m.Q = np.ascontiguousarray(Qc, dtype=np.float32)
        dQc, cQ, loss = O.bpr_shard_local(m, u, ci, cj, lr_t)
        grads = _a2a(dQc, cnt, srv_cnt, d, torch.float32)                          # back: one row per (peer, requested row)
        g, touched = np.zeros_like(self.item.Q), np.zeros(len(self.item.Q), np.int32)
        if self.planner is not None:                                               # the owner's index as the library builds it
            order, urows, useg = self.planner.owner_index(srv_rows, srv_cnt)
            for q, row in enumerate(urows):
                acc = np.zeros(d, np.float32)
                for k in range(useg[q], useg[q + 1]):                              # slot (= peer) order
                    acc = acc + grads[order[k]]
                g[row], touched[row] = acc, 1
            srv_cnt = []
        at = 0
        for p in range(len(srv_cnt)):                                              # peer order
            rows = srv_rows[at:at + int(srv_cnt[p])]
            first = touched[rows] == 0
            g[rows[first]] = grads[at:at + int(srv_cnt[p])][first]                 # 0 + x == x exactly
            g[rows[~first]] = g[rows[~first]] + grads[at:at + int(srv_cnt[p])][~first]
            touched[rows] = 1
            at += int(srv_cnt[p])
        O.adam_rows_lazy(self.item, g, touched, lr_t)
        lt = torch.tensor([loss], dtype=torch.float64)
        dist.all_reduce(lt)
        return float(lt)
