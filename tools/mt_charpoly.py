#!/usr/bin/env python3
"""Where csrc/mt_jump.cpp's PHI_LOW comes from: the characteristic polynomial of the MT19937 state transition, recovered from the
generator's own output by Berlekamp-Massey over GF(2) (Python integers as bit vectors), and checked against the raw word stream.

    python tools/mt_charpoly.py        # prints the 135 exponents (the highest is 19937)
"""
import numpy as np

N, M = 624, 397


def twist_block(mt):
    mt = mt.copy()
    for k in range(N):
        y = (int(mt[k]) & 0x80000000) | (int(mt[(k + 1) % N]) & 0x7fffffff)
        mt[k] = int(mt[(k + M) % N]) ^ (y >> 1) ^ (0x9908b0df if y & 1 else 0)
    return mt


def seed_key(s):
    mt = np.zeros(N, np.uint64)
    mt[0] = s
    for k in range(1, N):
        mt[k] = (1812433253 * (int(mt[k - 1]) ^ (int(mt[k - 1]) >> 30)) + k) & 0xffffffff
    return mt


def berlekamp_massey(bits):
    C, B, L, m, W = 1, 1, 0, 1, 0  # W: bit i = s[n - 1 - i]
    for n, b in enumerate(bits):
        d = b ^ (bin((C >> 1) & W).count("1") & 1)
        if d:
            T = C
            C ^= B << m
            if 2 * L <= n:
                L, B, m = n + 1 - L, T, 1
            else:
                m += 1
        else:
            m += 1
        W = (W << 1) | b
    return C, L


if __name__ == "__main__":
    blocks = [seed_key(12345)]
    for _ in range(69):
        blocks.append(twist_block(blocks[-1]))
    X = np.concatenate(blocks)
    C, L = berlekamp_massey([int(x) & 1 for x in X[N:N + 2 * 19937 + 100]])
    assert L == 19937
    exps = [e for e in range(L + 1) if (C >> (L - e)) & 1]  # phi(t): coefficient of t^e = c_(L-e)
    for n0 in (1, 5, N, N + 3):  # the raw stream satisfies it on full words from X[1] on
        acc = 0
        for e in exps:
            acc ^= int(X[n0 + e])
        assert acc == 0
    print(len(exps), "terms")
    print(", ".join(map(str, exps)))
