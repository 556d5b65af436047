#!/usr/bin/env python3
"""S-big shaped determinism check of the lazy step (heavy rows, cross-XCD ticket finish): the same 60 steps run three
times from the same state must leave bit-identical tables and losses.  Uses a 2 M x 200 k table to stay quick."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from cleverrec_amd import _lib, runtime
    sess = runtime.Session(0)
    dev = sess.device
    U, I, d, B, steps = 2_000_000, 200_000, 128, 6144, 60
    g = torch.Generator(device=dev).manual_seed(0)
    P0 = torch.empty((U, d), dtype=torch.float32, device=dev).normal_(0, 0.01, generator=g)
    Q0 = torch.empty((I, d), dtype=torch.float32, device=dev).normal_(0, 0.01, generator=g)
    n = steps * B
    u = torch.randint(0, U, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    w = 1.0 / torch.arange(1, I + 1, dtype=torch.float64, device=dev)
    i = torch.searchsorted(torch.cumsum(w / w.sum(), 0), torch.rand(n, generator=g, device=dev, dtype=torch.float64)).clamp_(max=I - 1).to(torch.int32)
    j = torch.randint(0, I, (n,), generator=g, device=dev, dtype=torch.int64).to(torch.int32)
    ref = None
    for rep in range(3):
        st = runtime.BprState.__new__(runtime.BprState)
        st.sess = sess
        st.P, st.Q = P0.clone(), Q0.clone()
        st.mP, st.vP, st.mQ, st.vQ = (torch.zeros_like(t) for t in (st.P, st.P, st.Q, st.Q))
        st.tables = _lib.BprTables(*(t.data_ptr() for t in (st.P, st.Q, st.mP, st.vP, st.mQ, st.vQ)), U, I, d)
        st.adam = _lib.Adam(1e-3, 0.9, 0.999, 1e-8, 0.01, _lib.ADAM_LAZY, 0.9, 0.999, 0)
        loss = st.train_epoch(u, i, j, B)
        torch.cuda.synchronize()
        cur = [loss.clone(), st.Q.clone(), st.mQ.clone(), st.vQ.clone(), st.P[: 1 << 18].clone()]
        if ref is None:
            ref = cur
        else:
            same = all(torch.equal(a, b) for a, b in zip(ref, cur))
            print("run %d bit-identical to run 0: %s" % (rep, same))
            if not same:
                sys.exit(1)
    top = torch.bincount(torch.cat([i[:B], j[:B]]).long(), minlength=I).max().item()
    print("ok (heaviest item row of the first batch: %d slots)" % top)


if __name__ == "__main__":
    main()
