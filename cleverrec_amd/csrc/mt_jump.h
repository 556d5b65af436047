// Jump-ahead for MT19937: what lets the device produce one NumPy-compatible word stream on many chains at once.
//
// The generator's state transition is linear over GF(2) with a primitive characteristic polynomial phi of degree 19937
// (Matsumoto & Nishimura 1998; the 135 non-zero terms are listed in mt_jump.cpp and checked against the recurrence
// when the table is built).  With g_J(t) = t^J mod phi, the state J steps ahead is g_J(A) s = XOR over the set bits i of
// g_J of the state i steps ahead -- and those states are just the generator's own raw word stream, so
//     X[J + 624 + w] = XOR_{i : bit i of g_J}  X[i + 624 + w]            (w = 0 .. 623, full 32-bit words)
// where X is the raw (untempered) stream that starts with the 624 key words.  (Offset 624: the low 31 bits of X[0] are
// not part of the 19937-bit state, every later word is a linear function of it.)
//
// A chain c of a super-block starts at block c*LB (a block = 624 words = one refill of the generator): it needs
// g_J for J = (c*LB - 1)*624, c = 1 .. T.  Chain T's block is the key of the next super-block.
#pragma once
#include <cstdint>

namespace crec_mtj {
constexpr int MT_N = 624, MT_M = 397, DEG = 19937;
constexpr int POLY_WORDS = 624;      // uint32 words per polynomial (19968 bits, the top 31 always zero)
constexpr int LB = 128;              // blocks per chain
constexpr int T = 128;               // chains per super-block (T*LB*624 = 10.2 M words)
constexpr int BASE_BLOCKS = 34;      // blocks of base stream a jump reads: words [624, 624 + DEG + 624)

// T polynomials, POLY_WORDS words each: entry c-1 = t^((c*LB-1)*624) mod phi.  Built once per process (threads),
// self-checked against plain generation; nullptr + *err on failure.
const uint32_t *jump_table(const char **err);
// host: one refill of the generator (key -> next block), and the jumped block of chain c (1 <= c <= T) by the formula above
void next_block(const uint32_t in[MT_N], uint32_t out[MT_N]);
bool jumped_block(const uint32_t key[MT_N], int c, uint32_t out[MT_N], const char **err);
} // namespace crec_mtj
