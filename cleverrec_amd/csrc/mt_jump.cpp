// Jump polynomials of MT19937 (see mt_jump.h): GF(2)[t] arithmetic modulo the generator's characteristic polynomial,
// the table of chain-start polynomials, and the host form of the jump (tests; the device form is in mt_sampler.hip).
#include "mt_jump.h"

#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace crec_mtj {
namespace {

constexpr int W = 312;      // 64-bit words of a reduced polynomial
constexpr int PW = 2 * W + 2; // words of an unreduced product

// phi(t) = t^19937 + sum of t^e over these exponents: the characteristic polynomial of the MT19937 transition
// (derived from the generator's own output by Berlekamp-Massey, tools/mt_charpoly.py; build_table() checks that the
// raw stream satisfies it before anything is built on it)
const int PHI_LOW[134] = {
    0,     1189,  1416,  1585,  1643,  1870,  2493,  2773,  3000,  3227,  3454,  3681,  3908,  4135,  4362,  4753,  5661,
    6337,  6569,  7129,  7477,  7525,  7583,  7752,  7979,  8206,  9505,  9901,  9969,  10128, 10693, 10761, 10920, 11089,
    11147, 11157, 11215, 11321, 11374, 11384, 11485, 11611, 11712, 11717, 11838, 11881, 11944, 11997, 12277, 12335, 12393,
    12504, 12509, 12620, 12673, 12731, 12736, 12789, 12905, 12958, 12963, 13137, 13185, 13190, 13243, 13301, 13412, 13528,
    13533, 13639, 13697, 13760, 13813, 13866, 14093, 14151, 14209, 14320, 14325, 14436, 14547, 14552, 14605, 14721, 14774,
    14779, 14953, 15001, 15006, 15059, 15117, 15228, 15344, 15349, 15455, 15513, 15576, 15629, 15682, 15909, 15967, 16025,
    16136, 16141, 16252, 16363, 16368, 16421, 16537, 16590, 16595, 16817, 16822, 16875, 16933, 17044, 17160, 17271, 17329,
    17445, 17498, 17725, 17783, 17841, 17952, 18068, 18179, 18237, 18406, 18633, 18691, 18860, 19087, 19314};

struct Poly {
    uint64_t w[W];
};

// p (PW words) modulo phi, in place: the part above t^19937 is folded down through phi's low terms until none is left
void reduce(uint64_t *p)
{
    constexpr int ws = DEG / 64, bs = DEG % 64;
    constexpr int HN = PW - ws; // words of the high part
    uint64_t h[HN + 1];
    for (;;) {
        bool any = false;
        for (int j = 0; j < HN; ++j) {
            const uint64_t lo = p[ws + j] >> bs;
            const uint64_t hi = (ws + j + 1 < PW) ? (p[ws + j + 1] << (64 - bs)) : 0;
            h[j] = lo | hi;
            any |= h[j] != 0;
        }
        if (!any)
            return;
        p[ws] &= (((uint64_t)1) << bs) - 1;
        for (int j = ws + 1; j < PW; ++j)
            p[j] = 0;
        int hn = HN;
        while (hn > 0 && h[hn - 1] == 0)
            --hn;
        for (int e : PHI_LOW) {
            const int w2 = e / 64, b2 = e % 64;
            if (b2 == 0) {
                for (int j = 0; j < hn; ++j)
                    p[w2 + j] ^= h[j];
            } else {
                for (int j = 0; j < hn; ++j) {
                    p[w2 + j] ^= h[j] << b2;
                    p[w2 + j + 1] ^= h[j] >> (64 - b2);
                }
            }
        }
    }
}

void modmul(const Poly &a, const Poly &b, Poly &out)
{
    // b shifted by 0..63 bits once, then one aligned XOR of W+1 words per set bit of a
    std::vector<uint64_t> sh((size_t)64 * (W + 1));
    for (int k = 0; k < 64; ++k) {
        uint64_t *s = &sh[(size_t)k * (W + 1)];
        for (int j = 0; j <= W; ++j) {
            const uint64_t cur = j < W ? b.w[j] : 0, prev = j > 0 ? b.w[j - 1] : 0;
            s[j] = k ? ((cur << k) | (prev >> (64 - k))) : cur;
        }
    }
    uint64_t prod[PW];
    std::memset(prod, 0, sizeof(prod));
    for (int i = 0; i < W; ++i) {
        uint64_t bits = a.w[i];
        while (bits) {
            const int k = __builtin_ctzll(bits);
            bits &= bits - 1;
            const uint64_t *s = &sh[(size_t)k * (W + 1)];
            uint64_t *d = prod + i;
            for (int j = 0; j <= W; ++j)
                d[j] ^= s[j];
        }
    }
    reduce(prod);
    std::memcpy(out.w, prod, sizeof(out.w));
}

Poly powmod(const Poly &base, unsigned e)
{
    Poly r, b = base, t;
    std::memset(r.w, 0, sizeof(r.w));
    r.w[0] = 1;
    while (e) {
        if (e & 1) {
            modmul(r, b, t);
            r = t;
        }
        e >>= 1;
        if (e) {
            modmul(b, b, t);
            b = t;
        }
    }
    return r;
}

inline uint32_t twist(uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return c ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}

std::vector<uint32_t> g_table; // T * POLY_WORDS
std::once_flag g_once;
const char *g_err = nullptr;

void apply_poly(const uint32_t *poly, const uint32_t *base /*BASE_BLOCKS * MT_N words*/, uint32_t out[MT_N])
{
    std::memset(out, 0, sizeof(uint32_t) * MT_N);
    for (int wi = 0; wi < POLY_WORDS; ++wi) {
        uint32_t bits = poly[wi];
        while (bits) {
            const int k = __builtin_ctz(bits);
            bits &= bits - 1;
            const uint32_t *x = base + MT_N + 32 * wi + k;
            for (int w = 0; w < MT_N; ++w)
                out[w] ^= x[w];
        }
    }
}

void base_stream(const uint32_t key[MT_N], std::vector<uint32_t> &x)
{
    x.resize((size_t)BASE_BLOCKS * MT_N);
    std::memcpy(x.data(), key, sizeof(uint32_t) * MT_N);
    for (int b = 1; b < BASE_BLOCKS; ++b)
        next_block(&x[(size_t)(b - 1) * MT_N], &x[(size_t)b * MT_N]);
}

void build_table()
{
    // the stream of init_genrand(5489): does it satisfy phi?  (n = 1 and n = 624: full words from X[1] on)
    uint32_t key[MT_N];
    key[0] = 5489u;
    for (int k = 1; k < MT_N; ++k)
        key[k] = 1812433253u * (key[k - 1] ^ (key[k - 1] >> 30)) + (uint32_t)k;
    std::vector<uint32_t> x;
    base_stream(key, x);
    for (int n : {1, MT_N, MT_N + 5}) {
        uint32_t acc = x[(size_t)n + DEG];
        for (int e : PHI_LOW)
            acc ^= x[(size_t)n + e];
        if (acc != 0) {
            g_err = "MT19937 characteristic polynomial does not annihilate the generator's stream";
            return;
        }
    }
    std::vector<Poly> P((size_t)T + 1);
    Poly m, S, t;
    std::memset(m.w, 0, sizeof(m.w));
    m.w[MT_N / 64] = (uint64_t)1 << (MT_N % 64); // t^624: one block
    P[1] = powmod(m, LB - 1);
    S = powmod(m, LB);
    unsigned hw = std::thread::hardware_concurrency();
    const int n_thr = (int)(hw < 1 ? 1 : (hw > 16 ? 16 : hw));
    for (int have = 1; have < T; have *= 2) { // P[c + have] = P[c] * S, S = t^(have * LB * 624)
        const int todo = (T - have) < have ? (T - have) : have;
        std::vector<std::thread> pool;
        auto work = [&](int lo, int hi) {
            for (int c = lo; c < hi; ++c)
                modmul(P[(size_t)c], S, P[(size_t)c + have]);
        };
        const int per = (todo + n_thr - 1) / n_thr;
        for (int lo = 1; lo <= todo; lo += per) {
            const int hi = (lo + per) < (todo + 1) ? (lo + per) : (todo + 1);
            try {
                pool.emplace_back(work, lo, hi);
            } catch (...) {
                work(lo, hi); // no more threads: do it here
            }
        }
        for (auto &th : pool)
            th.join();
        if (have * 2 < T) {
            modmul(S, S, t);
            S = t;
        }
    }
    g_table.resize((size_t)T * POLY_WORDS);
    for (int c = 1; c <= T; ++c)
        std::memcpy(&g_table[(size_t)(c - 1) * POLY_WORDS], P[(size_t)c].w, sizeof(uint32_t) * POLY_WORDS); // little-endian words
    // self-check against plain generation: chain 1 and chain 3 of the seed-5489 stream
    std::vector<uint32_t> blk(key, key + MT_N), nxt(MT_N);
    uint32_t got[MT_N];
    for (int b = 1; b <= 3 * LB; ++b) {
        next_block(blk.data(), nxt.data());
        blk.swap(nxt);
        if (b == LB || b == 3 * LB) {
            apply_poly(&g_table[(size_t)(b / LB - 1) * POLY_WORDS], x.data(), got);
            if (std::memcmp(got, blk.data(), sizeof(got)) != 0) {
                g_err = "MT19937 jump polynomial self-check failed";
                g_table.clear();
                return;
            }
        }
    }
}

} // namespace

void next_block(const uint32_t in[MT_N], uint32_t out[MT_N])
{
    int k = 0;
    for (; k < MT_N - MT_M; ++k)
        out[k] = twist(in[k], in[k + 1], in[k + MT_M]);
    for (; k < MT_N - 1; ++k)
        out[k] = twist(in[k], in[k + 1], out[k + MT_M - MT_N]);
    out[MT_N - 1] = twist(in[MT_N - 1], out[0], out[MT_M - 1]);
}

const uint32_t *jump_table(const char **err)
{
    try {
        std::call_once(g_once, build_table);
    } catch (...) {
        g_err = "out of memory building the MT19937 jump table";
    }
    if (g_table.empty()) {
        if (err)
            *err = g_err ? g_err : "MT19937 jump table unavailable";
        return nullptr;
    }
    return g_table.data();
}

bool jumped_block(const uint32_t key[MT_N], int c, uint32_t out[MT_N], const char **err)
{
    const uint32_t *tab = jump_table(err);
    if (!tab)
        return false;
    if (c < 1 || c > T) {
        if (err)
            *err = "chain index out of range";
        return false;
    }
    std::vector<uint32_t> x;
    base_stream(key, x);
    apply_poly(tab + (size_t)(c - 1) * POLY_WORDS, x.data(), out);
    return true;
}

} // namespace crec_mtj

extern "C" int crec_debug_mt_jump(const uint32_t key[624], int c, uint32_t out_block[624])
{
    if (!key || !out_block)
        return -1;
    const char *err = nullptr;
    return crec_mtj::jumped_block(key, c, out_block, &err) ? 0 : -1;
}
