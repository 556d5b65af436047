#!/usr/bin/env python3
"""NumPy model of the DEVICE form of the bit-exact epoch samplers (csrc/mt_sampler.hip), tile for tile.

Not product code and not the oracle: this is the algorithm the HIP kernels implement, written with one array element per
GPU thread so that every cut, prefix sum and hand-over of the kernels can be checked on a CPU against the streams the
reference itself produced (tests/golden/sampler_*.npz; tests/test_dsampler_model.py).  Reference: utils/sampler.py:10-99,
NumPy legacy randint / permutation.

Phases:
  0  raw MT19937 words (the device: jump-ahead chains; here: numpy's bit generator on the same state)
  1a candidates = raw words that pass randint's masked rejection (a pure filter: prefix-sum compaction)
  1b one workgroup walks the users in order: a tile of TILE candidates is tested against the user's seen-bitmap, an
     exclusive prefix sum gives every accepted candidate its slot, the tile is cut where the user's need is met or at the
     first repeat inside a positive's group (that candidate is consumed and rejected)
  2  swap targets of the legacy shuffle: tile of TILE raw words, accept iff (w & mask) <= i - (#accepted before it in the
     tile); elements whose decision depends on the unknown count are refined with prefix bounds until none is left; a tile
     is cut where the mask changes (i crosses a power of two)
  3  the permutation from the targets without executing the swaps: steps sorted by target, first-hop / head pointers,
     short pointer chases; then the gather of the output columns
"""
import numpy as np


def mask_of(m):
    m = int(m)
    m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16
    return m


def raw_words(key, pos, n):
    bg = np.random.MT19937()
    st = bg.state
    st["state"]["key"] = np.asarray(key, dtype=np.uint32)
    st["state"]["pos"] = int(pos)
    bg.state = st
    return bg.random_raw(n).astype(np.uint32)


def negatives(raw, ptr, items, item_nums, nr, TILE=1024):
    """-> (neg [n_pos*nr], raw words consumed)."""
    mx = item_nums - 1
    mask = mask_of(mx)
    v = raw & np.uint32(mask)
    ok = v <= mx
    cand, cand_raw = v[ok].astype(np.int64), np.nonzero(ok)[0]  # 1a
    n_users = len(ptr) - 1
    neg = np.full(int(ptr[-1]) * nr, -1, np.int64)
    pos = 0  # candidate index
    last = -1  # last candidate consumed
    for k in range(n_users):
        need = int(ptr[k + 1] - ptr[k]) * nr
        if need == 0:
            continue
        seen = np.zeros(item_nums, bool)
        seen[items[ptr[k]:ptr[k + 1]]] = True
        base = int(ptr[k]) * nr
        done = 0
        carry = np.zeros(max(nr - 1, 1), np.int64)  # last nr-1 accepted of this user (for groups that straddle tiles)
        while done < need:
            c = cand[pos:pos + TILE]
            if len(c) == 0:
                raise RuntimeError("out of candidates")
            f = ~seen[c]
            p = np.cumsum(f) - f  # exclusive prefix: rank among the accepted
            slot = done + p
            q = slot % nr
            D = np.concatenate([carry[-(nr - 1):] if nr > 1 else carry[:0], c[f]])  # D[(nr-1) + r]
            off = nr - 1
            dup = np.zeros(len(c), bool)
            for t in np.nonzero(f)[0]:
                if slot[t] >= need:
                    break
                r = p[t]
                for jj in range(1, q[t] + 1):
                    if D[off + r - jj] == c[t]:
                        dup[t] = True
            rem = need - done
            t_need = np.nonzero(f & (p == rem - 1))[0]
            t_need = int(t_need[0]) if len(t_need) else None
            t_dup = np.nonzero(dup)[0]
            t_dup = int(t_dup[0]) if len(t_dup) else None
            if t_dup is not None and (t_need is None or t_dup <= t_need):
                cut, acc = t_dup + 1, int(p[t_dup])  # [0, t_dup) kept, the repeat consumed and rejected
            elif t_need is not None:
                cut, acc = t_need + 1, rem
            else:
                cut, acc = len(c), int(f.sum())
            tk = np.nonzero(f[:cut] & (p[:cut] < acc))[0]
            neg[base + slot[tk]] = c[tk]
            if nr > 1 and acc:
                hist = np.concatenate([carry, c[tk]])
                carry = hist[-(nr - 1):] if len(hist) >= nr - 1 else np.concatenate([np.zeros(nr - 1 - len(hist), np.int64), hist])
            done += acc
            last = pos + cut - 1
            pos += cut
    used = int(cand_raw[last]) + 1 if last >= 0 else 0
    return neg, used


def shuffle_targets(raw, n, TILE=1024):
    """raw: words from the first one the shuffle may use -> (target[n] (target[0] unused), words consumed)."""
    tgt = np.zeros(n, np.int64)
    i, w = n - 1, 0
    while i >= 1:
        mask = mask_of(i)
        lo = (mask >> 1) + 1  # smallest i with this mask
        room = i - lo + 1  # steps left in this mask regime
        v = (raw[w:w + TILE] & np.uint32(mask)).astype(np.int64)
        T = len(v)
        if T == 0:
            raise RuntimeError("out of words")
        t = np.arange(T)
        acc = v <= i - t  # sure
        rej = v > i
        amb = ~acc & ~rej
        while amb.any():
            L = np.cumsum(acc) - acc
            U = L + np.cumsum(amb) - amb
            na = amb & (v <= i - U)
            nrj = amb & (v > i - L)
            acc |= na
            rej |= nrj
            amb &= ~(na | nrj)
        A = np.cumsum(acc) - acc
        tot = int(acc.sum())
        if tot >= room:  # the mask changes inside the tile: cut right behind the room-th accepted (what follows is judged under the next mask)
            e = int(np.nonzero(acc & (A == room - 1))[0][0]) + 1
            tot = room
        else:
            e = T
        sel = np.nonzero(acc[:e])[0]
        tgt[i - A[sel]] = v[sel]
        i -= tot
        w += e
    return tgt, w


def permutation_from_targets(tg):
    n = len(tg)
    NONE = -1
    idx = np.arange(n)
    key = tg.copy()
    if n:
        key[0] = n
    key[idx == tg] = n
    order = np.argsort(key, kind="stable")
    skey, sval = key[order], idx[order]
    head = np.full(n + 1, NONE)
    nxt = np.full(n, NONE)
    first = np.ones(n, bool)
    first[1:] = skey[1:] != skey[:-1]
    valid = skey < n
    head[skey[first & valid]] = sval[first & valid]
    same_next = np.zeros(n, bool)
    same_next[:-1] = skey[1:] == skey[:-1]
    m = valid & same_next
    nxt[sval[m]] = sval[np.nonzero(m)[0] + 1]
    start = np.where((idx == 0) | (tg == idx), idx, np.where(nxt != NONE, nxt, NONE))
    out = np.where(start == NONE, tg, 0)
    x = start.copy()
    live = x != NONE
    while True:
        h = np.where(live, head[np.where(live, x, 0)], NONE)
        mv = live & (h != NONE)
        if not mv.any():
            break
        x = np.where(mv, h, x)
    out = np.where(live, x, out)
    return out


def sample(kind, key, pos, keys, ptr, items, item_nums, nr, TILE=1024, budget=None):
    """-> (columns..., words consumed)"""
    n_pos = int(ptr[-1])
    n = {"pairwise": n_pos * nr, "pointwise": n_pos * (1 + nr), "cml": n_pos}[kind]
    budget = budget or (8 * (n_pos * nr + n) + 4096)
    raw = raw_words(key, pos, budget)
    neg, used1 = negatives(raw, ptr, items, item_nums, nr, TILE)
    tg, used2 = shuffle_targets(raw[used1:], n, TILE)
    perm = permutation_from_targets(tg)
    eu = np.repeat(np.asarray(keys, np.int64), np.diff(ptr))
    if kind == "pairwise":
        e = perm // nr
        cols = (eu[e], items[e].astype(np.int64), neg[perm])
    elif kind == "pointwise":
        w = 1 + nr
        e, t = perm // w, perm % w
        cols = (eu[e], np.where(t == 0, items[e], neg[e * nr + np.maximum(t - 1, 0)]), (t == 0).astype(np.float64))
    else:
        cols = (eu[perm], items[perm].astype(np.int64), neg.reshape(n_pos, nr)[perm])
    return cols, used1 + used2
