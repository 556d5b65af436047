"""CPU restatement (test infrastructure, like the rest of oracle/) of the index the lazy-family BPR steps read: TF1's bookkeeping for
the IndexedSlices gradients of one batch -- `unique` + `unsorted_segment_sum` over the batch's user ids and over its 2B item ids
(the aggregation behind model/ranking/BPR.py:31-44 when AdamOptimizer applies a gather's gradient) -- in the form csrc/bpr_train.hip
(bpr_index_batches, lazy branch) and csrc/index_big.hip write it and csrc/lazy_desc.h packs it.  Pure numpy, one batch at a time;
tests/test_gpu_bigbatch.py holds both device builders to it bit for bit.

Per batch of nb triples (u, i, j):
  user side   the nb (row, slot) pairs sorted by row, equal rows in slot order -> every slot's sorted position k, its row's
              segment [start, start + cnt)
  item side   the 2 nb pairs (slots 0..nb-1: the positives i, nb..2nb-1: the negatives j) sorted the same way; a row with two or
              more POSITIVE slots is a front row (ci = their number); front rows' positive slots take the first processing slots,
              row after row in sorted order; behind them the positions that touch another multi-occurrence item row, then the
              rest, each group in batch order
  descriptors int32[4] = {row | single << 31, k | hmod << 20, start | (ci & 0xfff) << 20, cnt | (ci >> 12) << 21}, hmod = (the
              front row's first processing slot) % spw."""
import numpy as np


def pack(row, single, k, start, cnt, ci, hmod):
    x = np.uint32(row) | (np.uint32(0x80000000) if single else np.uint32(0))
    y = np.uint32(k) | (np.uint32(hmod) << np.uint32(20))
    z = np.uint32(start) | ((np.uint32(ci) & np.uint32(0xfff)) << np.uint32(20))
    w = np.uint32(cnt) | ((np.uint32(ci) >> np.uint32(12)) << np.uint32(21))
    return np.array([x, y, z, w], dtype=np.uint32).view(np.int32)


def _runs(rows_sorted):
    """-> per sorted position: (start, cnt) of its run"""
    n = len(rows_sorted)
    heads = np.flatnonzero(np.r_[True, rows_sorted[1:] != rows_sorted[:-1]])
    ends = np.r_[heads[1:], n]
    start = np.repeat(heads, ends - heads)
    cnt = np.repeat(ends - heads, ends - heads)
    return start, cnt, heads, ends


def batch_index(u, i, j, spw):
    """one batch -> (u_desc [nb,4] by position, i_desc [nb,4], j_desc [nb,4], slot_pu [nb,2] by processing slot, (nb, n_front))"""
    u, i, j = (np.asarray(a, dtype=np.int64) for a in (u, i, j))
    nb = len(u)
    u_desc = np.zeros((nb, 4), np.int32)
    order_u = np.argsort(u, kind="stable")
    su = u[order_u]
    start, cnt, _, _ = _runs(su)
    for k in range(nb):
        u_desc[order_u[k]] = pack(su[k], cnt[k] == 1, k, start[k], cnt[k], 0, 0)
    rows = np.r_[i, j]
    order_q = np.argsort(rows, kind="stable")  # positives (slots < nb) before negatives inside a row, each in batch order
    sq = rows[order_q]
    start, cnt, heads, ends = _runs(sq)
    ci_run = np.array([int((order_q[h:e] < nb).sum()) for h, e in zip(heads, ends)])
    ci_run = np.where(ci_run >= 2, ci_run, 0)
    first_run = np.r_[0, np.cumsum(ci_run)[:-1]]
    n_front = int(ci_run.sum())
    ci = np.repeat(ci_run, ends - heads)
    first = np.repeat(first_run, ends - heads)
    i_tmp, j_tmp = np.zeros((nb, 4), np.int32), np.zeros((nb, 4), np.int32)
    order = np.full(nb, -1, np.int64)
    wt_i, wt_j = np.zeros(nb, np.int64), np.zeros(nb, np.int64)
    for k in range(2 * nb):
        slot = int(order_q[k])
        front = ci[k] >= 2
        d = pack(sq[k], cnt[k] == 1, k, start[k], cnt[k], ci[k], first[k] % spw if front else 0)
        if slot < nb:
            i_tmp[slot] = d
            if front:
                order[first[k] + (k - start[k])] = slot
            wt_i[slot] = -1 if front else cnt[k]
        else:
            j_tmp[slot - nb] = d
            wt_j[slot - nb] = cnt[k]
    rest = np.flatnonzero(wt_i != -1)
    multi = rest[np.maximum(wt_i[rest], wt_j[rest]) >= 2]
    single = rest[np.maximum(wt_i[rest], wt_j[rest]) < 2]
    order[n_front:n_front + len(multi)] = multi
    order[n_front + len(multi):] = single
    assert (np.sort(order) == np.arange(nb)).all()
    slot_pu = np.stack([order, u[order]], 1).astype(np.int32)
    return u_desc, i_tmp[order], j_tmp[order], slot_pu, (nb, n_front)
