// The bit-exact epoch samplers ON the device: utils/sampler.py:10-99 (pointwise / pairwise / CML) drawn from NumPy's
// legacy MT19937 stream, word for word, by kernels.  Same index streams and the same generator state afterwards as the
// reference (tests/golden/sampler_*.npz, tests/test_gpu_dsampler.py); the host form (crec_sampler.cpp) remains for SBPR
// and for shapes outside this file's limits.
//
// The reference's loop looks serial -- which stream word a draw takes depends on every earlier accept / reject -- but it is a
// filter, a prefix sum or a gather almost everywhere, and the two real chains have structure:
//
//  0  raw words.  MT19937 is linear over GF(2): the state J words ahead is a fixed XOR-combination (jump polynomial,
//     mt_jump.h) of the next 19937 states, i.e. of the stream itself.  `mt_base` writes 34 blocks from the input key,
//     `mt_chains` gives each of up to 128 workgroups its own start block by that combination (LDS, ~10 K XORs per word)
//     and lets it generate 128 blocks of 624 words: 10.2 M words per super-block, super-blocks chained through one
//     more jump.  Words stay raw (untempered) in HBM: the final generator state is 624 of them.
//  1a randint(item_nums) is a masked rejection (`v = word & mask; take v if v <= item_nums - 1`): a pure filter of the
//     stream -> `cand_*` kernels compact the survivors (count, scan, scatter), keeping each one's stream position.
//  1b the negatives.  User k+1's first candidate is wherever user k's last slot was filled -- but that place is predictable to
//     within a few hundred candidates, so `neg_tab` (a workgroup per user, all at once) tabulates the user's end for EVERY
//     start in its window, `neg_compose` / `neg_chain_starts` chain the tables by look-ups and `neg_emit` writes the negatives
//     from the true starts.  `neg_walk` (four waves, tiles of 1024 candidates through LDS rings) is the serial form: it takes
//     what the tables report instead of handling, and catalogues beyond their limits (`neg_chain_general` beyond 65 536 items).
//  2  np.random.permutation(n) is arange + for i = n-1..1: swap(a[i], a[draw in [0,i]]), the draw again a masked rejection but
//     with a bound that falls with every accepted word.  While i stays between two powers of two the accepted count is bounded
//     from both sides and every word whose two thresholds agree is decided, round after round over all words of the regime
//     (`perm_round`, `perm_count`, `perm_emit`); the steps below 65 536 are walked by one workgroup (`perm_targets`).
//  3  the permutation itself without executing a swap: position i ends up with what sat at its target before step i,
//     which is what the most recent earlier step with the same target put there ...: steps sorted by target (rocPRIM
//     radix sort), one pointer per step to the next step with its target, one per position to the first step that
//     targets it, and chains of expected length O(log n) (`perm_links`, `perm_gather`).  The gather of the output
//     columns (np.array(x)[s_idx], sampler.py:38,69,97) is fused into the chase.
//
// The generator state after the call (NumPy hands it to whoever draws next) is block b of the raw stream and a position.
#include "crec_internal.h"
#include "mt_jump.h"

#include <rocprim/rocprim.hpp>

#include <cmath>
#include <new>

namespace {

using crec_mtj::MT_M;
using crec_mtj::MT_N;

constexpr int TILE = 1024;       // threads of the two chain kernels = candidates / words per tile
constexpr int CBUF = 8192;       // LDS ring of candidates (neg_chain)
constexpr int UCACHE = 1024;     // users whose CSR bounds sit in LDS at a time
constexpr int CAND_BLK = 2048;   // raw words per workgroup of the compaction kernels (256 threads x 8)
constexpr int32_t NONE = -1;

enum { DS_OK = 0, DS_OUT_OF_CANDIDATES = 1, DS_OUT_OF_WORDS = 2, DS_INTERNAL = 3, DS_NEED_WALK = 4, DS_NEED_SERIAL_SHUFFLE = 5 };

struct DsStatus {
    int32_t status;
    int32_t pos_out;     // generator position after the call (0..624)
    int64_t n_cand;      // candidates the compaction produced
    int64_t raw_next1;   // stream index (in X) of the first word phase 2 may use
    int64_t raw_next2;   // ... of the first word nobody used: the generator stands here afterwards
    int64_t cand_used;
    int64_t perm_w;      // shuffle: the stream index the next regime / the walk starts at
    int32_t why, why_at; // DS_NEED_WALK: what the parallel negatives phase met (1 events, 2 knot of repeats, 3 start outside a window,
                         // 4 range too short, 5 a user's end does not check) and at which user / segment
    int64_t dbg[6];      // tile iterations, shader clocks, 100 MHz ticks of neg_chain and of perm_targets (crec_dsampler_stats)
    uint32_t key_out[MT_N];
};

__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b, uint32_t c)
{
    const uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return c ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

__host__ __device__ inline uint32_t mask_of(uint32_t m)
{
    m |= m >> 1, m |= m >> 2, m |= m >> 4, m |= m >> 8, m |= m >> 16;
    return m;
}

// one refill of the generator, old -> nw (both LDS, 624 words), by the threads [0, nt) of the workgroup, nt >= 227.
// new[k] needs old[k], old[k+1] and old[k+397] or new[k-227]: three rounds of <= 227 independent words.
__device__ __forceinline__ void twist_block(const uint32_t *old, uint32_t *nw, int tid)
{
    if (tid < MT_N - MT_M)
        nw[tid] = mt_twist(old[tid], old[tid + 1], old[tid + MT_M]);
    __syncthreads();
    if (tid < MT_N - MT_M) {
        const int k = tid + (MT_N - MT_M);
        nw[k] = mt_twist(old[k], old[k + 1], nw[k - (MT_N - MT_M)]);
    }
    __syncthreads();
    if (tid < MT_N - 2 * (MT_N - MT_M)) { // k = 454 .. 623
        const int k = tid + 2 * (MT_N - MT_M);
        nw[k] = k < MT_N - 1 ? mt_twist(old[k], old[k + 1], nw[k - (MT_N - MT_M)]) : mt_twist(old[MT_N - 1], nw[0], nw[MT_M - 1]);
    }
    __syncthreads();
}

// ---- phase 0 -----------------------------------------------------------------------------------------------------
// base[0 .. n_blocks*624): the raw stream that starts with the key
__global__ __launch_bounds__(256) void mt_base(const uint32_t *__restrict__ key, uint32_t *__restrict__ base, int n_blocks)
{
    __shared__ uint32_t buf[2][MT_N];
    const int tid = threadIdx.x;
    for (int k = tid; k < MT_N; k += 256)
        base[k] = buf[0][k] = key[k];
    __syncthreads();
    for (int b = 1; b < n_blocks; ++b) {
        const uint32_t *old = buf[(b - 1) & 1];
        uint32_t *nw = buf[b & 1];
        twist_block(old, nw, tid);
        for (int k = tid; k < MT_N; k += 256)
            base[(size_t)b * MT_N + k] = nw[k];
    }
}

constexpr int CHAIN_THREADS = 640; // >= 624: one thread per word of the jumped block
// workgroup c of the grid: chain c of the super-block.  Chain 0 starts from the key (block 0 of `base`); chain c > 0 from the
// block jumped to by polynomial c-1.  Chain `n_chains` (if next_key != nullptr) only jumps: its block is the next super-block's key.
// x: where this super-block's blocks go; blocks_total: how many of them exist (the last chain may be short).
__global__ __launch_bounds__(CHAIN_THREADS) void mt_chains(const uint32_t *__restrict__ base, const uint32_t *__restrict__ polys,
                                                           uint32_t *__restrict__ x, int64_t blocks_total, int n_chains,
                                                           uint32_t *__restrict__ next_key)
{
    extern __shared__ uint32_t lds[];
    constexpr int SPAN = crec_mtj::DEG + MT_N; // words [624, 624 + SPAN) of the base stream
    uint32_t *stream = lds;                   // SPAN words
    uint32_t *blk = lds + SPAN;               // 2 x 624
    uint16_t *bits = (uint16_t *)(blk + 2 * MT_N); // set bits of the polynomial, <= DEG entries
    __shared__ int n_bits_s;
    __shared__ int wcount[CHAIN_THREADS / 64];
    const int tid = threadIdx.x, c = blockIdx.x;
    const int64_t first = (int64_t)c * crec_mtj::LB;
    if (c == 0) {
        for (int k = tid; k < MT_N; k += CHAIN_THREADS)
            blk[k] = base[k];
        __syncthreads();
    } else {
        for (int k = tid; k < SPAN; k += CHAIN_THREADS)
            stream[k] = base[MT_N + k];
        // the polynomial's set bits as a list: thread t owns word t
        const uint32_t pw = tid < crec_mtj::POLY_WORDS ? polys[(size_t)(c - 1) * crec_mtj::POLY_WORDS + tid] : 0u;
        const int mine = __popc(pw);
        // exclusive scan of `mine` over the workgroup
        int incl = mine;
        const int lane = tid & 63, wave = tid >> 6;
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o);
            if (lane >= o)
                incl += t;
        }
        if (lane == 63)
            wcount[wave] = incl;
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wave; ++w)
            off += wcount[w];
        if (tid == CHAIN_THREADS - 1)
            n_bits_s = off + incl;
        int at = off + incl - mine;
        uint32_t m = pw;
        while (m) {
            const int k = __ffs(m) - 1;
            m &= m - 1;
            bits[at++] = (uint16_t)(32 * tid + k);
        }
        __syncthreads();
        const int nb = n_bits_s;
        if (tid < MT_N) {
            uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            int j = 0;
            for (; j + 4 <= nb; j += 4) { // four independent LDS reads in flight
                a0 ^= stream[bits[j] + tid];
                a1 ^= stream[bits[j + 1] + tid];
                a2 ^= stream[bits[j + 2] + tid];
                a3 ^= stream[bits[j + 3] + tid];
            }
            for (; j < nb; ++j)
                a0 ^= stream[bits[j] + tid];
            blk[tid] = a0 ^ a1 ^ a2 ^ a3;
        }
        __syncthreads();
    }
    if (c == n_chains) { // the hand-over block only
        if (tid < MT_N)
            next_key[tid] = blk[tid];
        return;
    }
    int64_t todo = blocks_total - first;
    todo = todo > crec_mtj::LB ? crec_mtj::LB : todo;
    for (int64_t b = 0; b < todo; ++b) {
        uint32_t *cur = blk + (b & 1) * MT_N;
        if (tid < MT_N)
            x[(size_t)(first + b) * MT_N + tid] = cur[tid];
        if (b + 1 < todo)
            twist_block(cur, blk + ((b + 1) & 1) * MT_N, tid);
    }
}

// ---- block-wide helpers of the chain kernels (1024 threads = 16 waves) ------------------------------------------------
__device__ __forceinline__ uint32_t lanes_below(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// exclusive prefix of a per-thread value packed per wave by the caller: `wave_total` (the wave's sum, same in all its lanes).
// wsum: LDS [16], 16-byte aligned; alternate between two arrays from call to call.  After the barrier every thread reads all
// sixteen sums (four broadcast reads, one LDS round trip) and adds up the ones below its wave -- a chain of cross-lane
// shuffles here cost six dependent LDS-crossbar trips per call, and these kernels are nothing but such chains.
__device__ __forceinline__ void block_offsets(uint32_t wave_total, uint32_t *wsum, int lane, int wave, uint32_t &before_wave, uint32_t &total)
{
    if (lane == 0)
        wsum[wave] = wave_total;
    __syncthreads();
    const uint4 *q = (const uint4 *)wsum;
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    const uint32_t v[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    uint32_t bef = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        tot += v[w];
        bef += w < wave ? v[w] : 0u;
    }
    before_wave = bef;
    total = tot;
}

// min over the workgroup of one value per wave (the wave's lanes all pass the same value); wmin as wsum above
__device__ __forceinline__ uint32_t read_min16(const uint32_t *w16)
{
    const uint4 *q = (const uint4 *)w16;
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    const uint32_t v[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
    uint32_t m = 0xffffffffu;
#pragma unroll
    for (int w = 0; w < 16; ++w)
        m = v[w] < m ? v[w] : m;
    return m;
}

__device__ __forceinline__ uint32_t block_min(uint32_t wave_val, uint32_t *wmin, int lane, int wave)
{
    if (lane == 0)
        wmin[wave] = wave_val;
    __syncthreads();
    return read_min16(wmin);
}

// ---- LDS-DMA plumbing of the two walks ---------------------------------------------------------------------------------
// A walk's inputs (stream words, candidates) are consumed strictly in order, a tile per trip round the loop, and a trip is a
// few hundred cycles -- shorter than one HBM / Infinity-Cache round trip.  A register prefetch does not help (the compiler
// waits for the register before the loop closes: the first version spent 1.4-2 us per trip waiting for its loads), so the
// tiles are brought into an LDS ring by LDS-DMA (`global_load_lds_dword`: no register destination), several tiles ahead,
// and retired by a counted `s_waitcnt vmcnt(N)` in front of a raw `s_barrier`; the ring is read one phase later
// (cdna_hip_programming.md, "Read a staged buffer one phase AFTER the wait that retires it").  The statements are inline asm,
// so the compiler neither counts nor drains them; nothing else in the loops loads from global memory.
__device__ __forceinline__ void glds_dword(const void *gsrc, uint32_t lds_byte_addr)
{
    unsigned keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr); // M0: the wave's 256 destination bytes start here
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)p; }
template <int N> __device__ __forceinline__ void wait_vm_then_barrier()
{
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- phase 1a: candidates ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cand_count(const uint32_t *__restrict__ x, int64_t r0, int64_t n_words, uint32_t mask, uint32_t mx,
                                                  uint32_t *__restrict__ counts)
{
    __shared__ uint32_t ws[4];
    const int64_t at = (int64_t)blockIdx.x * CAND_BLK + (int64_t)threadIdx.x * 8;
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (at + k < n_words)
            n += (mt_temper(x[r0 + at + k]) & mask) <= mx;
    for (int o = 32; o >= 1; o >>= 1)
        n += __shfl_xor(n, o);
    if ((threadIdx.x & 63) == 0)
        ws[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0)
        counts[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// counts -> exclusive offsets (in place), total -> st->n_cand.  One workgroup of 1024 threads.
__global__ __launch_bounds__(1024) void cand_scan(uint32_t *__restrict__ counts, int n_blocks, DsStatus *__restrict__ st)
{
    __shared__ __align__(16) uint32_t wsum[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint64_t carry = 0;
    int par = 0;
    for (int b0 = 0; b0 < n_blocks; b0 += 1024, par ^= 1) {
        const uint32_t v = b0 + tid < n_blocks ? counts[b0 + tid] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(incl, o);
            if (lane >= o)
                incl += t;
        }
        uint32_t before, total;
        block_offsets(__shfl(incl, 63), wsum[par], lane, wave, before, total);
        if (b0 + tid < n_blocks)
            counts[b0 + tid] = (uint32_t)(carry + before + incl - v); // < 2^32: a compaction never covers more words than that
        carry += total;
    }
    if (tid == 0)
        st->n_cand = (int64_t)carry;
}

__global__ __launch_bounds__(256) void cand_scatter(const uint32_t *__restrict__ x, int64_t r0, int64_t n_words, uint32_t mask, uint32_t mx,
                                                    const uint32_t *__restrict__ offs, int32_t *__restrict__ cand, uint32_t *__restrict__ cand_raw)
{
    __shared__ uint32_t ws[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t at = (int64_t)blockIdx.x * CAND_BLK + (int64_t)tid * 8;
    uint32_t v[8];
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        v[k] = 0xffffffffu;
        if (at + k < n_words) {
            const uint32_t t = mt_temper(x[r0 + at + k]) & mask;
            if (t <= mx) {
                v[k] = t;
                ++n;
            }
        }
    }
    uint32_t incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= o)
            incl += t;
    }
    if (lane == 63)
        ws[wave] = incl;
    __syncthreads();
    uint32_t off = offs[blockIdx.x] + incl - n;
    for (int w = 0; w < wave; ++w)
        off += ws[w];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (v[k] != 0xffffffffu) {
            cand[off] = (int32_t)v[k];
            cand_raw[off] = (uint32_t)(r0 + at + k);
            ++off;
        }
}

// ---- the users' bitmaps of training items (once per data set) -----------------------------------------------------------
__global__ __launch_bounds__(256) void build_bitmaps(const int64_t *__restrict__ ptr, const int32_t *__restrict__ items, int bw,
                                                     uint32_t *__restrict__ bm)
{
    const int64_t k = blockIdx.x;
    for (int64_t e = ptr[k] + threadIdx.x; e < ptr[k + 1]; e += 256) {
        const int32_t it = items[e];
        atomicOr(&bm[k * bw + (it >> 5)], 1u << (it & 31));
    }
}

// ---- phase 1b ------------------------------------------------------------------------------------------------------
struct NegArgs {
    const int32_t *cand;
    const uint32_t *cand_raw;
    const int64_t *ptr;
    int64_t n_users;
    const uint32_t *bitmaps;
    int bw, nbuf, nr;
    int32_t *neg;
    DsStatus *st;
    int64_t raw_begin;
    uint32_t cand_cap; // elements allocated behind `cand` (a multiple of 4): the walk's 16-byte requests stay inside
    uint32_t neg_spare; // index of a spare element behind `neg`: where a lane without a result stores
};

// PWORDS: bitmap words a thread moves per user (bw <= PWORDS * 1024)
template <int PWORDS> __global__ __launch_bounds__(TILE) void neg_chain_general(NegArgs a)
{
    extern __shared__ uint32_t lds[];
    uint32_t *bm = lds;                                     // nbuf x bw
    int32_t *cbuf = (int32_t *)(bm + (((size_t)a.nbuf * a.bw + 1) & ~(size_t)1)); // CBUF
    int32_t *D = cbuf + CBUF;                               // (nr - 1) + TILE: the accepted of a tile behind the last nr-1 of the tiles before
    int64_t *uptr = (int64_t *)(D + ((a.nr - 1 + TILE + 1) & ~1)); // UCACHE + 1 (8-byte aligned: the sizes before it are even)
    __shared__ __align__(16) uint32_t wsum[2][16], wmin[2][16], wneed[2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nr = a.nr, bw = a.bw, nbuf = a.nbuf, CARRY = a.nr - 1;
    const int64_t n_cand = a.st->n_cand;

    int64_t filled = 0;
    for (int f = 0; f < 3; ++f) {
        const int64_t idx = filled + tid;
        cbuf[idx & (CBUF - 1)] = idx < n_cand ? a.cand[idx] : 0;
        filled += TILE;
    }
    int64_t kc0 = 0;
    for (int t = tid; t <= UCACHE; t += TILE)
        uptr[t] = kc0 + t <= a.n_users ? a.ptr[kc0 + t] : 0;
    for (int s = 0; s < nbuf; ++s)
        if (s < a.n_users)
#pragma unroll
            for (int j = 0; j < PWORDS; ++j) {
                const int w = tid + j * TILE;
                if (w < bw)
                    bm[(size_t)s * bw + w] = a.bitmaps[(size_t)s * bw + w];
            }
    __syncthreads();

    uint32_t pend[PWORDS];
    int pend_slot = -1;
    int64_t pos = 0, last = -1, iters = 0;
    const uint64_t clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    int par = 0;
    bool failed = false;
    for (int64_t k = 0; k < a.n_users && !failed; ++k) {
        if (k - kc0 >= UCACHE) {
            __syncthreads();
            kc0 = k;
            for (int t = tid; t <= UCACHE; t += TILE)
                uptr[t] = kc0 + t <= a.n_users ? a.ptr[kc0 + t] : 0;
            __syncthreads();
        }
        const int64_t p0 = uptr[k - kc0], p1 = uptr[k - kc0 + 1];
        const int64_t need = (p1 - p0) * nr, base = p0 * nr;
        const uint32_t *ubm = bm + (size_t)(k % nbuf) * bw;
        int64_t done = 0;
        while (done < need) {
            if (pos >= n_cand) { // the compaction was sized too small for this epoch's rejections: the host doubles it and starts over
                failed = true;
                break;
            }
            const bool do_fill = filled - pos < 3 * TILE && filled < n_cand;
            int32_t fillv = 0;
            if (do_fill) {
                const int64_t idx = filled + tid;
                fillv = idx < n_cand ? a.cand[idx] : 0;
            }
            const int64_t gi = pos + tid;
            const bool valid = gi < n_cand;
            const int32_t c = cbuf[gi & (CBUF - 1)];
            const bool f = valid && !((ubm[c >> 5] >> (c & 31)) & 1u);
            const uint64_t mf = __ballot(f);
            uint32_t before, total;
            block_offsets((uint32_t)__popcll(mf), wsum[par], lane, wave, before, total);
            const uint32_t p = before + lanes_below(mf);
            const int64_t rem = need - done; // >= 1
            const uint32_t dmod = (uint32_t)(done % nr);
            bool dup = false;
            if (nr > 1) {
                if (f)
                    D[CARRY + p] = c;
                __syncthreads();
                if (f && (int64_t)p < rem) {
                    const int q = (int)((dmod + p) % (uint32_t)nr); // earlier members of this positive's group
                    for (int jj = 1; jj <= q; ++jj)
                        dup |= D[CARRY + (int)p - jj] == c;
                }
            }
            // first repeat (with its rank) and the thread that holds the user's last slot
            const uint64_t md = __ballot(dup), mn = __ballot(f && (int64_t)p == rem - 1);
            uint32_t wd = 0xffffffffu, wn = 0xffffffffu;
            if (md) {
                const int l = __ffsll((unsigned long long)md) - 1;
                wd = ((uint32_t)(wave * 64 + l) << 11) | __shfl(p, l);
            }
            if (mn)
                wn = (uint32_t)(wave * 64 + __ffsll((unsigned long long)mn) - 1);
            if (lane == 0)
                wneed[par][wave] = wn;
            const uint32_t dmin = block_min(wd, wmin[par], lane, wave); // its barrier also covers wneed
            const uint32_t t_need = read_min16(wneed[par]);
            const uint32_t t_dup = dmin == 0xffffffffu ? 0xffffffffu : (dmin >> 11);
            int cut;
            int64_t acc;
            if (t_dup != 0xffffffffu && t_dup <= t_need) { // [0, t_dup) kept; the repeat is consumed and rejected
                cut = (int)t_dup + 1;
                acc = dmin & 2047u;
            } else if (t_need != 0xffffffffu) {
                cut = (int)t_need + 1;
                acc = rem;
            } else {
                const int64_t left = n_cand - pos;
                cut = left < TILE ? (int)left : TILE;
                acc = total;
            }
            if (f && tid < cut && (int64_t)p < acc)
                a.neg[base + done + p] = c;
            if (nr > 1 && done + acc < need) { // the group in progress may reach back into this tile: keep its last nr-1 accepted
                int32_t v = 0;
                if (tid < CARRY)
                    v = D[acc + tid];
                __syncthreads();
                if (tid < CARRY)
                    D[tid] = v;
            }
            if (do_fill) {
                cbuf[(filled + tid) & (CBUF - 1)] = fillv;
                filled += TILE;
            }
            done += acc;
            last = pos + cut - 1;
            pos += cut;
            par ^= 1;
            ++iters;
        }
        // user k is done: its ring slot takes user k + nbuf.  The words loaded at the previous switch are stored now.
        __syncthreads();
        if (pend_slot >= 0)
#pragma unroll
            for (int j = 0; j < PWORDS; ++j) {
                const int w = tid + j * TILE;
                if (w < bw)
                    bm[(size_t)pend_slot * bw + w] = pend[j];
            }
        pend_slot = -1;
        if (k + nbuf < a.n_users) {
#pragma unroll
            for (int j = 0; j < PWORDS; ++j) {
                const int w = tid + j * TILE;
                pend[j] = w < bw ? a.bitmaps[(size_t)(k + nbuf) * bw + w] : 0u;
            }
            pend_slot = (int)(k % nbuf);
        }
    }
    if (tid == 0) {
        a.st->dbg[0] = iters;
        a.st->dbg[1] = (int64_t)(__builtin_amdgcn_s_memtime() - clk0);
        a.st->dbg[2] = (int64_t)(__builtin_amdgcn_s_memrealtime() - rt0);
        a.st->cand_used = pos;
        a.st->raw_next1 = last >= 0 ? (int64_t)a.cand_raw[last] + 1 : a.raw_begin;
        if (failed)
            a.st->status = DS_OUT_OF_CANDIDATES;
    }
}

// ---- the walks as they run wherever their limits allow: FOUR waves, four tile elements per lane ---------------------------------
// A trip round a walk's loop is one link of a serial chain, and what a trip costs is the instructions of its longest wave:
// the first version (1024 threads, one element each) put four waves on every SIMD, each of them executing the trip's uniform
// bookkeeping -- prefix sums, the cut, the state, the ring management, ~500 instructions -- and measured 3 100 (shuffle
// targets) and 5 000-9 000 (negatives) cycles per tile with the loads already hidden.  With one wave per SIMD and four
// elements per lane the bookkeeping runs once per SIMD instead of four times and the element work keeps all lanes busy.
// Element (wave w, row j, lane l) sits at tile offset w * 256 + j * 64 + l: group g = 4 w + j, sixteen groups in tile order.
constexpr int WT = 256;             // threads of a walk
constexpr int NWAIT = 5;            // fills (LDS-DMA requests of 4 KiB) that may be in flight across a barrier
// what a wave may leave outstanding at a walk's counted wait, in vector-memory operations: a trip issues exactly four stores (one per
// element row, unconditionally: rows without a result store to a dummy slot) and at most one fill, so the NOPS youngest operations
// span at least NWAIT trips and hold at most NWAIT fills
constexpr int NOPS = 5 * NWAIT;
constexpr int CRING = 16384;        // ring of candidates / stream words
constexpr int CDEPTH = NWAIT + 1;   // fills ahead of the tile being read
constexpr int WALK_NBUF = NWAIT + 3; // bitmap ring (see the wait's comment in neg_walk)
constexpr int WALK_BW_MAX = 2048;   // bitmap words per user the ring takes (item_nums <= 65536)
constexpr int WALK_NR_MAX = 256;

// 16 bytes per lane, 1 KiB per wave: LDS destination = lds_byte_addr + 16 * lane
__device__ __forceinline__ void glds_x4(const void *gsrc, uint32_t lds_byte_addr)
{
    unsigned keep;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}

// inclusive prefix sum inside every row of 16 lanes (DPP row shifts: no LDS, no cross-lane permutes)
__device__ __forceinline__ uint32_t row_scan16(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); // row_shr:1, lanes without a source add 0
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    return v;
}

// ---- phase 1b: the negatives of every positive, users in the reference's order ---------------------------------------------------
// Rules as in neg_chain_general above (a tile is cut behind the user's last slot or behind the first repeat inside a positive's
// group); candidates and the users' bitmaps arrive in LDS rings by LDS-DMA several tiles / users ahead.
template <int NRT> __global__ __launch_bounds__(WT) void neg_walk(NegArgs a)
{
    extern __shared__ uint32_t lds[];
    static_assert(WALK_NBUF == 8, "ring slot = user & 7");
    const int nr = NRT ? NRT : a.nr, CARRY = nr - 1, bw = a.bw;
    // ONE shared object: [bitmap ring][candidate ring][D: accepted of the tile behind the last nr-1 of the tiles before, one spare][counts][cuts][CSR bounds]
    uint32_t *bm = lds;
    int32_t *cbuf = (int32_t *)(bm + (size_t)WALK_NBUF * bw);
    int32_t *D = cbuf + CRING;
    uint32_t *cnt = (uint32_t *)(D + ((CARRY + TILE + 1 + 3) & ~3)); // [2][16]
    uint32_t *red = cnt + 32;                                        // [2][8]: per wave {first repeat | its rank, thread of the last slot}
    uint32_t *uptr = red + 16;                                       // [UCACHE + 1]: CSR bounds (< 2^31: checked by the host)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t n_cand = (uint32_t)a.st->n_cand, n_users = (uint32_t)a.n_users;
    const uint64_t clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    const uint32_t cbuf_at = lds_addr(cbuf), bm_at = lds_addr(bm);
    const uint32_t d_spare = (uint32_t)(CARRY + TILE); // where lanes without an accepted candidate put theirs

    uint32_t filled = 0;
    const auto fill = [&]() { // candidates [filled, filled + 1024) -> ring; one 16-byte request per lane
        uint32_t idx = filled + wave * 256 + lane * 4;
        idx = idx + 4 <= a.cand_cap ? idx : a.cand_cap - 4;
        glds_x4(a.cand + idx, cbuf_at + ((filled + wave * 256) & (CRING - 1)) * 4);
        filled += TILE;
    };
    // user u's bitmap -> ring slot u & 7, 256 words per wave and request.  Past the last user the request is still made (the
    // last user's words, into a slot nobody reads any more): the wait below counts on the requests that follow one
    const auto load_bitmap = [&](uint32_t u) {
        const uint32_t slot_at = bm_at + (u & 7u) * (uint32_t)bw * 4u;
        const uint32_t src = u < n_users ? u : n_users - 1;
        for (int w0 = wave * 256; w0 < bw; w0 += 1024)
            glds_x4(a.bitmaps + (size_t)src * bw + w0 + lane * 4, slot_at + (uint32_t)(w0 * 4));
    };
    const auto load_bounds = [&](uint32_t k0) {
        for (uint32_t t = tid; t <= UCACHE; t += WT)
            uptr[t] = k0 + t <= n_users ? (uint32_t)a.ptr[k0 + t] : 0u;
    };
    for (int f = 0; f <= CDEPTH; ++f)
        fill();
    for (uint32_t u = 0; u < WALK_NBUF; ++u)
        load_bitmap(u);
    uint32_t kc0 = 0;
    load_bounds(0);
    wait_vm_then_barrier<0>();

    uint32_t pos = 0, iters = 0;
    int64_t last = -1;
    int par = 0;
    bool failed = false;
    const uint32_t t0 = wave * 256 + lane;
    for (uint32_t k = 0; k < n_users && !failed; ++k) {
        if (k - kc0 >= UCACHE) { // (ordinary loads: drains the DMA queue; once per 1024 users)
            lds_barrier();
            kc0 = k;
            load_bounds(k);
            lds_barrier();
        }
        const uint32_t p0 = uptr[k - kc0], p1 = uptr[k - kc0 + 1];
        const uint32_t need = (p1 - p0) * (uint32_t)nr, base = p0 * (uint32_t)nr;
        const uint32_t *ubm = bm + (size_t)(k & 7u) * bw;
        uint32_t done = 0, dmod = 0;
        if (need == 0) // a user without rows makes no trip (no stores to count on): retire everything
            wait_vm_then_barrier<0>();
        while (done < need) {
            if (pos >= n_cand) { // the compaction was sized too small for this epoch's rejections: the host doubles it and starts over
                failed = true;
                break;
            }
            // one tile consumed per trip at most: with one fill per trip filled - pos >= (CDEPTH + 1) tiles, so the NEXT trip's
            // tile ends before the NWAIT fills that the wait below may leave in flight
            if (filled - pos < (CDEPTH + 1) * TILE && filled < n_cand)
                fill();
            // ---- candidates against the user's bitmap; per group: count of the accepted
            int32_t c[4];
            uint32_t pw[4], cw[4];
            uint64_t mf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
                c[j] = cbuf[(pos + t0 + j * 64) & (CRING - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t word = ubm[(uint32_t)c[j] >> 5];
                mf[j] = __ballot(pos + t0 + j * 64 < n_cand && !((word >> (c[j] & 31)) & 1u));
                cw[j] = (uint32_t)__popcll(mf[j]);
                pw[j] = lanes_below(mf[j]);
            }
            if (lane == 0)
                *(uint4 *)(cnt + par * 16 + wave * 4) = make_uint4(cw[0], cw[1], cw[2], cw[3]);
            // The wait: everything but this wave's NOPS youngest vector-memory operations has completed, which leaves at most NWAIT
            // fills in flight (the candidates read behind this barrier were requested CDEPTH fills ago).  A bitmap first read behind
            // it was requested nbuf users ago: the trips of the nbuf - 2 users since (four stores each) and their own bitmap
            // requests are more than NOPS younger operations.
            wait_vm_then_barrier<NOPS>();
            const uint32_t mine = lane < 16 ? cnt[par * 16 + lane] : 0u;
            const uint32_t incl = row_scan16(mine);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15);
            uint32_t p[4];
            bool f[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p[j] = (uint32_t)__builtin_amdgcn_readlane((int)(incl - mine), wave * 4 + j) + pw[j];
                f[j] = (mf[j] >> lane) & 1;
            }
            const uint32_t rem = need - done; // >= 1
            // ---- repeats inside a positive's group (sampler.py:24,59,88): the accepted of the tile in rank order (D, behind the
            // last nr-1 accepted of the tiles before), every one compared with the earlier members of its group.  No branches:
            // lanes without an accepted candidate write a spare slot and read slots they do not use.
            uint64_t md[4] = {0, 0, 0, 0};
            if (nr > 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    D[f[j] ? CARRY + p[j] : d_spare] = c[j];
                lds_barrier();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int32_t *at = D + CARRY + p[j];
                    uint32_t dup; // integer logic throughout: `&&` would make the reads conditional, i.e. branches
                    if (NRT == 4) {
                        const int32_t v1 = at[-1], v2 = at[-2], v3 = at[-3];
                        const uint32_t q = (dmod + p[j]) & 3u;
                        dup = ((uint32_t)(q >= 1) & (uint32_t)(v1 == c[j])) | ((uint32_t)(q >= 2) & (uint32_t)(v2 == c[j])) |
                              ((uint32_t)(q >= 3) & (uint32_t)(v3 == c[j]));
                    } else {
                        const uint32_t q = (dmod + p[j]) % (uint32_t)nr;
                        dup = 0;
                        for (int jj = 1; jj <= CARRY; ++jj) // uniform trip count; the reads beyond q are not used
                            dup |= (uint32_t)((uint32_t)jj <= q) & (uint32_t)(at[-jj] == c[j]);
                    }
                    md[j] = __ballot((dup & (uint32_t)f[j] & (uint32_t)(p[j] < rem)) != 0);
                }
            }
            // this wave's first repeat (thread << 11 | rank) and the thread that holds the user's last slot
            uint32_t wd = 0xffffffffu, wn = 0xffffffffu;
#pragma unroll
            for (int j = 3; j >= 0; --j) {
                if (md[j]) {
                    const int l = __ffsll((unsigned long long)md[j]) - 1;
                    wd = ((uint32_t)(wave * 256 + j * 64 + l) << 11) | (uint32_t)__builtin_amdgcn_readlane((int)p[j], l);
                }
                const uint64_t mn = __ballot(f[j] && p[j] == rem - 1);
                if (mn)
                    wn = (uint32_t)(wave * 256 + j * 64 + __ffsll((unsigned long long)mn) - 1);
            }
            if (lane == 0)
                *(uint2 *)(red + par * 8 + wave * 2) = make_uint2(wd, wn);
            lds_barrier();
            const uint4 ra = *(const uint4 *)(red + par * 8), rb = *(const uint4 *)(red + par * 8 + 4);
            const uint32_t dmin = min(min(ra.x, ra.z), min(rb.x, rb.z)), t_need = min(min(ra.y, ra.w), min(rb.y, rb.w));
            const uint32_t t_dup = dmin == 0xffffffffu ? 0xffffffffu : (dmin >> 11);
            uint32_t cut, acc;
            if (t_dup != 0xffffffffu && t_dup <= t_need) { // [0, t_dup) kept; the repeat is consumed and rejected
                cut = t_dup + 1;
                acc = dmin & 2047u;
            } else if (t_need != 0xffffffffu) {
                cut = t_need + 1;
                acc = rem;
            } else {
                const uint32_t left = n_cand - pos;
                cut = left < TILE ? left : TILE;
                acc = total;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { // exactly four stores per trip (see NOPS): lanes without a negative write the spare slot behind the array
                const bool ok = f[j] && t0 + j * 64 < cut && p[j] < acc;
                a.neg[ok ? base + done + p[j] : a.neg_spare] = c[j];
            }
            if (nr > 1 && done + acc < need) { // the group in progress may reach back into this tile: keep its last nr-1 accepted
                int32_t v = 0;
                if (tid < CARRY)
                    v = D[acc + tid];
                lds_barrier();
                if (tid < CARRY)
                    D[tid] = v;
            }
            done += acc;
            dmod = NRT == 4 ? ((dmod + acc) & 3u) : (dmod + acc) % (uint32_t)nr;
            last = (int64_t)pos + cut - 1;
            pos += cut;
            par ^= 1;
            ++iters;
        }
        load_bitmap(k + WALK_NBUF); // user k is done: its ring slot takes the user nbuf ahead
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // no LDS-DMA may be in flight when the workgroup ends
    if (tid == 0) {
        a.st->dbg[0] = iters;
        a.st->dbg[1] = (int64_t)(__builtin_amdgcn_s_memtime() - clk0);
        a.st->dbg[2] = (int64_t)(__builtin_amdgcn_s_memrealtime() - rt0);
        a.st->cand_used = pos;
        a.st->raw_next1 = last >= 0 ? (int64_t)a.cand_raw[last] + 1 : a.raw_begin;
        if (failed)
            a.st->status = DS_OUT_OF_CANDIDATES;
    }
}

// ---- phase 1b, the PARALLEL form: every user on its own workgroup, the chain reduced to table look-ups ----------------------------
// The walk above is a chain of ~8 000 dependent tiles on one CU (13 ms at ml-1m shape) because user k+1's first candidate is
// wherever user k's last slot was filled.  But where that is can be predicted: user k takes need_k / (1 - p_k) candidates on
// average (p_k = share of the catalogue it has seen) with a known variance, so its start lies within a few hundred candidates
// of a position known in advance (plan: lo_k, W_k = twice six standard deviations + 64).  So:
//   neg_tab    (one workgroup per user, all at once) tests the user's bitmap against every candidate it could possibly meet,
//              and writes, for EVERY start s in the window, where the user would end: T_k[s - lo_k] = E_k(s).  Without repeats
//              that is a rank look-up (the (rank(s) + need)-th accepted candidate); repeats inside a positive's group are found
//              as `events` -- two equal accepted candidates at most nr + 1 ranks apart, a property of the candidates alone --
//              and whether an event is a repeat for a given start (both in one group of nr) is decided per start.
//   neg_compose  one thread per (segment of 64 users, start in the segment's first window): 64 look-ups -> F_g
//   neg_chain_starts  one wave: ~100 look-ups through F give every segment's true start, then one thread per segment replays its
//              users' tables: start[k] for every user.  Nothing here is approximate: a look-up outside a window, a table entry
//              whose range was too short, or a knot of repeats the event logic does not cover is REPORTED (DS_NEED_WALK) and the
//              sample is redone by the walk.
//   neg_emit   (one workgroup per user) the user's negatives from its true start.
constexpr uint32_t T_INVALID = 0xffffffffu;
constexpr int TAB_EV = 1024;    // events per user beyond which the sample goes to the walk (tiny catalogues)
constexpr int TAB_SK = 256;     // repeats per user, likewise
constexpr int TAB_SEG = 64;     // users per segment
constexpr int TAB_NR_MAX = 30; // events look nr + 1 ranks back: one bit each in a 32-bit mask

struct TabArgs {
    const int32_t *cand;
    const uint32_t *bitmaps;
    int bw, nr;
    uint32_t n_users;
    const uint32_t *lo, *W, *len, *toff, *need, *base; // the plan, per user
    uint32_t *T;
    uint32_t *start; // [n_users + 1]
    DsStatus *st;
    uint32_t rcap, wcap;
    int32_t *neg;
    const uint32_t *cand_raw;
    int64_t raw_begin;
};

__device__ __forceinline__ uint32_t wave_scan64(uint32_t v, int lane) // inclusive prefix sum over the wave
{
    v = row_scan16(v);
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), r1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31),
                   r2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
    const int row = lane >> 4;
    return v + (row >= 1 ? r0 : 0u) + (row >= 2 ? r1 : 0u) + (row >= 3 ? r2 : 0u);
}

// exclusive prefix of v over the 256 threads and the total; ws: LDS [8], two alternating halves
__device__ __forceinline__ uint32_t block_scan256(uint32_t v, uint32_t *ws, int lane, int wave, uint32_t &total)
{
    const uint32_t incl = wave_scan64(v, lane);
    if (lane == 63)
        ws[wave] = incl;
    __syncthreads();
    const uint32_t w0 = ws[0], w1 = ws[1], w2 = ws[2], w3 = ws[3];
    total = w0 + w1 + w2 + w3;
    return incl - v + (wave >= 1 ? w0 : 0u) + (wave >= 2 ? w1 : 0u) + (wave >= 3 ? w2 : 0u);
}

// the accepted candidates of cand[c0 .. c0 + L) for the user whose bitmap is ubm (LDS): acc[r] = value << 16 | offset of the r-th
// one (r < rcap; item ids and offsets below 65 536 on this path), rkw[t] = accepted before offset t (t < Wn).  Returns their number.
__device__ uint32_t build_accepted(const int32_t *__restrict__ cand, uint32_t c0, uint32_t L, const uint32_t *ubm, uint32_t *acc, uint32_t rcap,
                                   uint16_t *rkw, uint32_t Wn, uint32_t *ws)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t cum = 0;
    int par = 0;
    for (uint32_t ch = 0; ch < L; ch += 1024, par ^= 4) {
        uint32_t f[4], sum = 0;
        int32_t c[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t t = ch + tid * 4 + j;
            c[j] = t < L ? cand[c0 + t] : 0;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t t = ch + tid * 4 + j;
            f[j] = t < L && !((ubm[(uint32_t)c[j] >> 5] >> (c[j] & 31)) & 1u);
            sum += f[j];
        }
        uint32_t total;
        uint32_t rb = cum + block_scan256(sum, ws + par, lane, wave, total);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t t = ch + tid * 4 + j;
            if (t < Wn)
                rkw[t] = (uint16_t)rb;
            if (f[j] && rb < rcap)
                acc[rb] = ((uint32_t)c[j] << 16) | t;
            rb += f[j];
        }
        cum += total;
    }
    __syncthreads();
    return cum;
}

// events: pairs of equal accepted candidates at most D ranks apart, as (r << 8 | d) in rank order (r the later one, d the distance).
// Returns the number of events (capped at TAB_EV + 1).
__device__ uint32_t build_events(const uint32_t *acc, uint32_t R, uint32_t D, uint32_t *ev, uint32_t *ws)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t nev = 0;
    int par = 0;
    for (uint32_t rc = 0; rc < R; rc += 1024, par ^= 4) {
        uint32_t cnt = 0, hit[4] = {0, 0, 0, 0}; // bit d of hit[j]: the candidate d ranks back equals this one (D <= 31 on this path)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t r = rc + tid * 4 + j;
            if (r < R) {
                const uint32_t mine = acc[r] >> 16;
                for (uint32_t d = 1; d <= D && d <= r; ++d)
                    hit[j] |= (uint32_t)((acc[r - d] >> 16) == mine) << d;
                cnt += __popc(hit[j]);
            }
        }
        uint32_t total;
        uint32_t at = nev + block_scan256(cnt, ws + par, lane, wave, total);
        if (total) // rare enough: a candidate that equals one close before it
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t r = rc + tid * 4 + j;
                for (uint32_t h = hit[j]; h; h &= h - 1) {
                    if (at < TAB_EV)
                        ev[at] = (r << 8) | (uint32_t)(__ffs(h) - 1);
                    ++at;
                }
            }
        nev += total;
        if (nev > TAB_EV)
            break;
    }
    __syncthreads();
    return nev > TAB_EV ? TAB_EV + 1 : nev;
}

// How many of the events are repeats when the user starts at accepted rank rs: walks the events in rank order with the number
// of repeats so far and the three latest repeated ranks.  An event (rp, r) is a repeat iff rp belongs to this start (rp >= rs),
// is itself no repeat, and lies in r's group: fewer taken candidates between them than the group holds before r.  Exact while
// at most two repeats fall inside any stretch of D ranks; anything denser is flagged (cplx) and handled by the walk.
// rec (LDS, may be null): the repeated ranks, written by lanes with `rec_writer`.
__device__ __forceinline__ uint32_t eval_skips(const uint32_t *ev, uint32_t nev, uint32_t rs, uint32_t need, uint32_t nr, uint32_t D, bool &cplx,
                                               uint32_t *rec, bool rec_writer)
{
    uint32_t skips = 0;
    int32_t sk1 = -1, sk2 = -1, sk3 = -1;
    for (uint32_t e = 0; e < nev; ++e) {
        const uint32_t r = ev[e] >> 8, d = ev[e] & 255u, rp = r - d;
        if (rp >= rs && r <= rs + need - 1 + skips && (int32_t)r != sk1) {
            if (sk3 >= 0 && sk3 >= (int32_t)r - (int32_t)D)
                cplx = true; // three repeats this close: more could hide outside what is tracked
            const uint32_t between = (r - rp - 1) - (uint32_t)(sk1 > (int32_t)rp) - (uint32_t)(sk2 > (int32_t)rp);
            const bool rp_repeat = sk1 == (int32_t)rp || sk2 == (int32_t)rp;
            const uint32_t q = ((r - rs) - skips) % nr;
            if (!rp_repeat && between < q) {
                sk3 = sk2, sk2 = sk1, sk1 = (int32_t)r;
                if (rec && rec_writer && skips < TAB_SK)
                    rec[skips] = r;
                ++skips;
                if (sk3 >= 0 && sk3 + (int32_t)(D + nr) >= sk1)
                    cplx = true;
            }
        }
    }
    return skips;
}

// LDS of neg_tab / neg_emit: [bitmap row bw][acc rcap][ev TAB_EV][rec TAB_SK][ws 8][rkw wcap x 16 bit]
// The users are launched in two classes (`ids`: the users of this launch): the many with short ranges share a CU several at a time,
// the few long ones get the LDS they need.
__global__ __launch_bounds__(256) void neg_tab(TabArgs a, const uint32_t *__restrict__ ids, uint32_t rcap, uint32_t wcap)
{
    extern __shared__ uint32_t lds[];
    uint32_t *ubm = lds, *acc = ubm + a.bw, *ev = acc + rcap, *ws = ev + TAB_EV + TAB_SK;
    uint16_t *rkw = (uint16_t *)(ws + 8);
    const int tid = threadIdx.x;
    const uint32_t k = ids[blockIdx.x], lo = a.lo[k], W = a.W[k], need = a.need[k];
    uint32_t *T = a.T + a.toff[k];
    const uint32_t n_cand = (uint32_t)a.st->n_cand;
    if (need == 0) { // no rows: the next user starts where this one would
        for (uint32_t i = tid; i < W; i += 256)
            T[i] = lo + i;
        return;
    }
    if (lo >= n_cand) {
        for (uint32_t i = tid; i < W; i += 256)
            T[i] = T_INVALID;
        return;
    }
    const uint32_t L = min(a.len[k], n_cand - lo);
    for (int w = tid; w < a.bw; w += 256)
        ubm[w] = a.bitmaps[(size_t)k * a.bw + w];
    __syncthreads();
    const uint32_t R = min(build_accepted(a.cand, lo, L, ubm, acc, rcap, rkw, min(W, L), ws), rcap);
    const uint32_t D = (uint32_t)a.nr + 1;
    const uint32_t nev = a.nr > 1 ? build_events(acc, R, D, ev, ws) : 0u;
    if (nev > TAB_EV) {
        if (tid == 0)
            a.st->status = DS_NEED_WALK, a.st->why = 1, a.st->why_at = (int32_t)k;
        return;
    }
    bool cplx = false;
    for (uint32_t i = tid; i < W; i += 256) {
        uint32_t E = T_INVALID;
        if (i < L) {
            const uint32_t rs = rkw[i];
            const uint32_t tr = rs + need - 1 + (nev ? eval_skips(ev, nev, rs, need, (uint32_t)a.nr, D, cplx, nullptr, false) : 0u);
            if (tr < R)
                E = lo + (acc[tr] & 0xffffu) + 1;
        }
        T[i] = E;
    }
    if (cplx)
        a.st->status = DS_NEED_WALK, a.st->why = 2, a.st->why_at = (int32_t)k;
}

// F[foff[g] + i] = where segment g ends when its first user starts at lo[first] + i
__global__ __launch_bounds__(256) void neg_compose(TabArgs a, const uint32_t *__restrict__ foff, uint32_t *__restrict__ F)
{
    const uint32_t g = blockIdx.x, u0 = g * TAB_SEG, u1 = min(u0 + TAB_SEG, a.n_users);
    const uint32_t i = blockIdx.y * 256 + threadIdx.x;
    if (i >= a.W[u0])
        return;
    uint32_t x = a.lo[u0] + i;
    for (uint32_t u = u0; u < u1; ++u) {
        const uint32_t idx = x - a.lo[u];
        if (idx >= a.W[u]) { // (also x < lo[u]: the difference wraps)
            x = T_INVALID;
            break;
        }
        x = a.T[a.toff[u] + idx];
        if (x == T_INVALID)
            break;
    }
    F[foff[g] + i] = x;
}

// one wave: the segments' true starts through F (lane 0), then every user's through T (a lane per segment)
constexpr uint32_t TAB_MAXSEG = 8192;
__global__ __launch_bounds__(64) void neg_chain_starts(TabArgs a, const uint32_t *__restrict__ foff, const uint32_t *__restrict__ F, uint32_t n_seg)
{
    __shared__ uint32_t seg_start[TAB_MAXSEG];
    const int lane = threadIdx.x;
    if (a.st->status != DS_OK)
        return;
    bool bad = false;
    if (lane == 0) {
        uint32_t x = 0;
        for (uint32_t g = 0; g < n_seg; ++g) {
            seg_start[g] = x;
            const uint32_t idx = x - a.lo[g * TAB_SEG];
            if (idx >= a.W[g * TAB_SEG]) {
                bad = true;
                a.st->why = 3, a.st->why_at = (int32_t)g;
                break;
            }
            x = F[foff[g] + idx];
            if (x == T_INVALID) {
                bad = true;
                a.st->why = 4, a.st->why_at = (int32_t)g;
                break;
            }
        }
        if (!bad) {
            a.start[a.n_users] = x;
            a.st->cand_used = x;
            a.st->raw_next1 = x > 0 ? (int64_t)a.cand_raw[x - 1] + 1 : a.raw_begin;
        }
    }
    if (__any(bad)) {
        if (lane == 0)
            a.st->status = DS_NEED_WALK;
        return;
    }
    __syncthreads();
    for (uint32_t g = lane; g < n_seg; g += 64) {
        uint32_t x = seg_start[g];
        const uint32_t u1 = min((g + 1) * TAB_SEG, a.n_users);
        for (uint32_t u = g * TAB_SEG; u < u1; ++u) {
            a.start[u] = x;
            x = a.T[a.toff[u] + (x - a.lo[u])]; // inside the window and valid: F said so
        }
    }
}

// the users of this launch by their TRUE ranges: ids of the users whose range (start[k+1] - start[k]) is at most / above `split`
// are not known to the host, so every user is tried by both launches and takes the one its range belongs to
__global__ __launch_bounds__(256) void neg_emit(TabArgs a, uint32_t rcap, uint32_t min_len, uint32_t max_len)
{
    extern __shared__ uint32_t lds[];
    uint32_t *ubm = lds, *acc = ubm + a.bw, *ev = acc + rcap, *rec = ev + TAB_EV, *ws = rec + TAB_SK;
    const int tid = threadIdx.x;
    if (a.st->status != DS_OK)
        return;
    const uint32_t k = blockIdx.x, need = a.need[k], base = a.base[k];
    if (need == 0)
        return;
    const uint32_t s0 = a.start[k], L = a.start[k + 1] - s0;
    if (L < min_len || L > max_len)
        return; // the other launch's user
    for (int w = tid; w < a.bw; w += 256)
        ubm[w] = a.bitmaps[(size_t)k * a.bw + w];
    __syncthreads();
    const uint32_t R = build_accepted(a.cand, s0, L, ubm, acc, rcap, nullptr, 0, ws);
    const uint32_t D = (uint32_t)a.nr + 1;
    const uint32_t nev = a.nr > 1 && R <= rcap ? build_events(acc, R, D, ev, ws) : 0u;
    bool cplx = R > rcap || nev > TAB_EV;
    uint32_t skips = 0;
    if (!cplx && nev)
        skips = eval_skips(ev, nev, 0u, need, (uint32_t)a.nr, D, cplx, rec, tid == 0);
    // the table said this user ends exactly here: its last candidate is its last slot
    if (cplx || skips > TAB_SK || R != need + skips || (acc[R - 1] & 0xffffu) != L - 1) {
        if (tid == 0)
            a.st->status = DS_NEED_WALK, a.st->why = 5, a.st->why_at = (int32_t)k;
        return;
    }
    __syncthreads();
    for (uint32_t r = tid; r < R; r += 256) {
        uint32_t before = 0;
        bool rep = false;
        for (uint32_t q = 0; q < skips; ++q) {
            before += rec[q] < r;
            rep |= rec[q] == r;
        }
        if (!rep)
            a.neg[base + r - before] = (int32_t)(acc[r] >> 16);
    }
}

// ---- phase 2 -------------------------------------------------------------------------------------------------------
struct PermLds {
    uint32_t ring[CRING];
    uint32_t cnt[2][16];
    int cut_at[2];
};

// tgt[i], i = n-1 .. 1: the swap target of step i of the legacy shuffle.  Also leaves the generator's final state.
// i0: the first step this walk takes (n - 1, or below the regimes that perm_round / perm_emit resolved in parallel: then the
// stream position comes from st->perm_w)
__global__ __launch_bounds__(WT) void perm_targets(const uint32_t *__restrict__ x, int64_t x_words64, int64_t i0, int after_regimes,
                                                   uint32_t *__restrict__ tgt, DsStatus *__restrict__ st)
{
    __shared__ __align__(16) PermLds L; // ONE shared object (a second one beside an LDS-DMA target can cost a vmcnt(0) per read)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (st->status != DS_OK)
        return;
    const uint32_t x_words = (uint32_t)x_words64;
    uint32_t w = (uint32_t)(after_regimes ? st->perm_w : st->raw_next1);
    int32_t i = (int32_t)i0;
    uint32_t iters = 0;
    int par = 0;
    bool failed = false;
    const uint64_t clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
    const uint32_t ring_at = lds_addr(L.ring);
    uint32_t filled = w & ~255u; // the ring holds stream words [.., filled); a wave's 256 words never wrap
    const auto fill = [&]() {
        uint32_t idx = filled + wave * 256 + lane * 4;
        idx = idx + 4 <= x_words ? idx : x_words - 4; // x_words is a multiple of 624: of 4
        glds_x4(x + idx, ring_at + ((filled + wave * 256) & (CRING - 1)) * 4);
        filled += TILE;
    };
    for (int f = 0; f <= CDEPTH + 1; ++f)
        fill();
    wait_vm_then_barrier<0>();
    const uint32_t t0 = wave * 256 + lane;
    while (i >= 1) {
        if (w >= x_words) {
            failed = true;
            break;
        }
        // one tile consumed per trip at most, so one fill per trip keeps filled - w >= CDEPTH + 1 tiles (+ the 255 words `filled`
        // may start below w): the next trip's tile ends before the NWAIT fills that the wait below may leave in flight
        if (filled - w < (CDEPTH + 1) * TILE + 256)
            fill();
        const uint32_t left = x_words - w;
        const uint32_t navail = left < TILE ? left : TILE;
        const uint32_t mask = mask_of((uint32_t)i);
        const int32_t room = i - (int32_t)(mask >> 1); // steps left before the mask changes: i .. (mask >> 1) + 1
        int32_t v[4];
        bool acc[4], amb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t t = t0 + j * 64;
            v[j] = (int32_t)(mt_temper(L.ring[(w + t) & (CRING - 1)]) & mask);
            const bool valid = t < navail;
            acc[j] = valid && v[j] <= i - (int32_t)t; // even if every word before it in the tile is accepted
            amb[j] = valid && !acc[j] && v[j] <= i;
        }
        uint32_t Lc[4] = {0, 0, 0, 0}, tot = 0;
        for (int round = 0;; ++round) {
            if (round > TILE + 2) { // every round settles at least the first undecided word: cannot happen
                failed = true;
                break;
            }
            uint64_t ma[4], mb[4];
            uint32_t pk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ma[j] = __ballot(acc[j]);
                mb[j] = __ballot(amb[j]);
                pk[j] = (uint32_t)__popcll(ma[j]) | ((uint32_t)__popcll(mb[j]) << 16);
            }
            if (lane == 0)
                *(uint4 *)(L.cnt[par] + wave * 4) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            wait_vm_then_barrier<NOPS>(); // at most NWAIT fills stay in flight: the tile of the NEXT trip has landed
            const uint32_t mine = lane < 16 ? L.cnt[par][lane] : 0u;
            const uint32_t incl = row_scan16(mine);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 15);
            par ^= 1;
            uint32_t bef[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bef[j] = (uint32_t)__builtin_amdgcn_readlane((int)(incl - mine), wave * 4 + j);
                Lc[j] = (bef[j] & 0xffffu) + lanes_below(ma[j]);
            }
            tot = total & 0xffffu;
            if ((total >> 16) == 0)
                break;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (amb[j]) {
                    const uint32_t U = Lc[j] + (bef[j] >> 16) + lanes_below(mb[j]);
                    if (v[j] <= i - (int32_t)U) {
                        acc[j] = true;
                        amb[j] = false;
                    } else if (v[j] > i - (int32_t)Lc[j]) {
                        amb[j] = false;
                    }
                }
        }
        if (failed)
            break;
        uint32_t e = navail;
        if ((int32_t)tot >= room) { // cut right behind the room-th accepted word: what follows is judged under the next mask
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (acc[j] && (int32_t)Lc[j] == room - 1)
                    L.cut_at[par] = (int)(t0 + j * 64 + 1);
            lds_barrier();
            e = (uint32_t)L.cut_at[par];
            tot = (uint32_t)room;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { // exactly four stores per trip (see NOPS): lanes without a target write tgt[0], which no step reads
            const bool ok = acc[j] && t0 + j * 64 < e && Lc[j] < tot;
            tgt[ok ? i - (int32_t)Lc[j] : 0] = (uint32_t)v[j];
        }
        i -= (int32_t)tot;
        w += e;
        ++iters;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // no LDS-DMA may be in flight when the workgroup ends
    if (failed) {
        if (tid == 0)
            st->status = w >= x_words ? DS_OUT_OF_WORDS : DS_INTERNAL;
        return;
    }
    // RandomState afterwards: `pos` words of block b consumed; a block that is used up stays (pos = 624) until the next draw
    const uint32_t b = (w % MT_N == 0 && w > 0) ? w / MT_N - 1 : w / MT_N;
    for (int t = tid; t < MT_N; t += WT)
        st->key_out[t] = x[(size_t)b * MT_N + t];
    if (tid == 0) {
        st->pos_out = (int32_t)(w - b * MT_N);
        st->raw_next2 = w;
        st->dbg[3] = iters;
        st->dbg[4] = (int64_t)(__builtin_amdgcn_s_memtime() - clk0);
        st->dbg[5] = (int64_t)(__builtin_amdgcn_s_memrealtime() - rt0);
    }
}

// ---- phase 2, the large steps in parallel --------------------------------------------------------------------------------------
// While i stays between two powers of two (a `regime`: one mask, R steps from hi down) word t of the stream is accepted iff
// (word & mask) <= hi - c_t, c_t = the accepted before it: a threshold that depends on the answer.  Instead of walking, bound
// c_t from both sides -- accepted-for-sure before t <= c_t <= that + undecided before t -- and decide every word whose two
// thresholds agree; the undecided thin out like t^n / (n! M^n) from round to round (M = mask + 1 >= 2 R), so eight to ten
// rounds settle a regime of a million words, each round one launch over all of its words (prefix sums of the round before, per
// workgroup of 2048 words).  The regimes themselves follow one another (the next one starts where this one's R-th accepted
// word lies); the steps below 65 536 -- 90 K words in all -- are left to the walk above.  Rounds that find nothing undecided
// return at once; a regime that is not settled after PR_ROUNDS, or runs out of words, is reported (the walk takes the sample).
constexpr int PR_BLK = 2048;
constexpr int PR_ROUNDS = 12;
constexpr int PR_MIN_STEP = 65536;
struct PermRegime {
    int32_t hi, R;
    uint32_t mask, tmax;
    int nb, first; // workgroups; first regime of the shuffle (its start is raw_next1)
};

__device__ __forceinline__ uint32_t regime_start(const DsStatus *st, const PermRegime &g) { return (uint32_t)(g.first ? st->raw_next1 : st->perm_w); }

// cls: 0 undecided, 1 accepted, 2 rejected or behind the regime's end.  blk: [PR_ROUNDS][2][nbmax] accepted / undecided per workgroup.
__global__ __launch_bounds__(256) void perm_round(const uint32_t *__restrict__ x, uint32_t x_words, PermRegime g, int r, uint8_t *__restrict__ cls,
                                                  uint32_t *__restrict__ blk, int nbmax, const DsStatus *__restrict__ st)
{
    __shared__ uint32_t ws[8], red[12];
    if (st->status != DS_OK)
        return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t b = blockIdx.x, w0 = regime_start(st, g);
    uint32_t L0 = 0, A0 = 0;
    if (r > 0) {
        const uint32_t *pa = blk + (size_t)((r - 1) * 2) * nbmax, *pb = pa + nbmax;
        uint32_t la = 0, lb = 0, all = 0;
        for (int q = tid; q < g.nb; q += 256) {
            const uint32_t va = pa[q], vb = pb[q];
            all += vb;
            if ((uint32_t)q < b)
                la += va, lb += vb;
        }
        for (int o = 32; o >= 1; o >>= 1)
            la += __shfl_xor(la, o), lb += __shfl_xor(lb, o), all += __shfl_xor(all, o);
        if (lane == 0)
            red[wave] = la, red[4 + wave] = lb, red[8 + wave] = all;
        __syncthreads();
        L0 = red[0] + red[1] + red[2] + red[3], A0 = red[4] + red[5] + red[6] + red[7];
        if (red[8] + red[9] + red[10] + red[11] == 0) { // the round before left nothing undecided: hand its counts on
            if (tid == 0)
                blk[(size_t)(r * 2) * nbmax + b] = pa[b], blk[(size_t)(r * 2 + 1) * nbmax + b] = 0;
            return;
        }
    }
    const uint32_t t0 = b * PR_BLK + tid * 8;
    int32_t v[8];
    uint32_t stt[8], sa = 0, sb = 0;
    uint64_t old = 0;
    if (r > 0)
        old = *(const uint64_t *)(cls + t0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t t = t0 + j;
        const bool valid = t < g.tmax && w0 + t < x_words;
        v[j] = valid ? (int32_t)(mt_temper(x[w0 + t]) & g.mask) : 0x7fffffff;
        if (r == 0)
            stt[j] = !valid || v[j] > g.hi ? 2u : (v[j] <= g.hi - (int32_t)t ? 1u : 0u); // at most t accepted before word t
        else
            stt[j] = (uint32_t)(old >> (8 * j)) & 255u;
        sa += stt[j] == 1u, sb += stt[j] == 0u;
    }
    uint32_t na = sa, nu = sb;
    if (r > 0) {
        uint32_t total;
        const uint32_t ex = block_scan256(sa | (sb << 16), ws, lane, wave, total);
        uint32_t Lt = L0 + (ex & 0xffffu), Ut = Lt + A0 + (ex >> 16);
        na = 0, nu = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t was = stt[j];
            if ((int32_t)Lt >= g.R)
                stt[j] = 2u; // behind the regime's last step
            else if (was == 0u) {
                if (v[j] <= g.hi - (int32_t)Ut)
                    stt[j] = 1u;
                else if (v[j] > g.hi - (int32_t)Lt)
                    stt[j] = 2u;
            }
            Lt += was == 1u, Ut += was != 2u;
            na += stt[j] == 1u, nu += stt[j] == 0u;
        }
    }
    uint64_t nw = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        nw |= (uint64_t)stt[j] << (8 * j);
    *(uint64_t *)(cls + t0) = nw;
    for (int o = 32; o >= 1; o >>= 1)
        na += __shfl_xor(na, o), nu += __shfl_xor(nu, o);
    __syncthreads();
    if (lane == 0)
        red[wave] = na, red[4 + wave] = nu;
    __syncthreads();
    if (tid == 0) {
        blk[(size_t)(r * 2) * nbmax + b] = red[0] + red[1] + red[2] + red[3];
        blk[(size_t)(r * 2 + 1) * nbmax + b] = red[4] + red[5] + red[6] + red[7];
    }
}

// after the rounds: per workgroup the accepted words (fin[b]) and whether anything is left undecided (-> the serial shuffle)
__global__ __launch_bounds__(256) void perm_count(PermRegime g, const uint8_t *__restrict__ cls, uint32_t *__restrict__ fin, DsStatus *__restrict__ st)
{
    __shared__ uint32_t red[8];
    if (st->status != DS_OK)
        return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t c8 = *(const uint64_t *)(cls + (size_t)blockIdx.x * PR_BLK + tid * 8);
    uint32_t na = 0, nu = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t c = (uint32_t)(c8 >> (8 * j)) & 255u;
        na += c == 1u, nu += c == 0u;
    }
    for (int o = 32; o >= 1; o >>= 1)
        na += __shfl_xor(na, o), nu += __shfl_xor(nu, o);
    if (lane == 0)
        red[wave] = na, red[4 + wave] = nu;
    __syncthreads();
    if (tid == 0) {
        fin[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
        if (red[4] + red[5] + red[6] + red[7])
            st->status = DS_NEED_SERIAL_SHUFFLE;
    }
}

// tgt[hi - rank] for the regime's accepted words; the word behind the R-th one is where the next regime starts
__global__ __launch_bounds__(256) void perm_emit(const uint32_t *__restrict__ x, PermRegime g, const uint8_t *__restrict__ cls, const uint32_t *__restrict__ fin,
                                                 uint32_t *__restrict__ tgt, DsStatus *__restrict__ st)
{
    __shared__ uint32_t ws[8], red[8];
    if (st->status != DS_OK)
        return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t b = blockIdx.x, w0 = regime_start(st, g);
    uint32_t la = 0, all = 0;
    for (int q = tid; q < g.nb; q += 256) {
        const uint32_t va = fin[q];
        all += va;
        if ((uint32_t)q < b)
            la += va;
    }
    for (int o = 32; o >= 1; o >>= 1)
        la += __shfl_xor(la, o), all += __shfl_xor(all, o);
    if (lane == 0)
        red[wave] = la, red[4 + wave] = all;
    __syncthreads();
    const uint32_t L0 = red[0] + red[1] + red[2] + red[3], total_acc = red[4] + red[5] + red[6] + red[7];
    if ((int32_t)total_acc < g.R) { // the words ran out before the regime's last step: a larger budget
        if (b == 0 && tid == 0)
            st->status = DS_OUT_OF_WORDS;
        return;
    }
    const uint32_t t0 = b * PR_BLK + tid * 8;
    const uint64_t c8 = *(const uint64_t *)(cls + t0);
    uint32_t sa = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        sa += ((uint32_t)(c8 >> (8 * j)) & 255u) == 1u;
    uint32_t total;
    uint32_t Lt = L0 + block_scan256(sa, ws, lane, wave, total);
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (((uint32_t)(c8 >> (8 * j)) & 255u) == 1u) {
            if ((int32_t)Lt < g.R) {
                tgt[g.hi - (int32_t)Lt] = mt_temper(x[w0 + t0 + j]) & g.mask;
                if ((int32_t)Lt == g.R - 1)
                    st->perm_w = (int64_t)w0 + t0 + j + 1;
            }
            ++Lt;
        }
}

// ---- phase 3 -------------------------------------------------------------------------------------------------------
// (phase 3 reads what the walks wrote: after a walk that ran out of stream -- the host redoes the sample with a larger budget --
// `tgt` is not fully written, and nothing may be indexed with it)
__global__ __launch_bounds__(256) void perm_keys(const uint32_t *__restrict__ tgt, int64_t n, uint32_t *__restrict__ key, uint32_t *__restrict__ val,
                                                 int32_t *__restrict__ head, const DsStatus *__restrict__ st)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || st->status != DS_OK)
        return;
    const uint32_t t = i == 0 ? (uint32_t)n : tgt[i];
    key[i] = (i == 0 || t == (uint32_t)i) ? (uint32_t)n : t; // step 0 does not exist; a self-swap moves nothing
    val[i] = (uint32_t)i;
    head[i] = NONE;
}

// sorted by (target, step): head[p] = first step that targets p, nxt[i] = the next step with step i's target
__global__ __launch_bounds__(256) void perm_links(const uint32_t *__restrict__ skey, const uint32_t *__restrict__ sval, int64_t n,
                                                  int32_t *__restrict__ head, int32_t *__restrict__ nxt, const DsStatus *__restrict__ st)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n || st->status != DS_OK)
        return;
    const uint32_t p = skey[q], i = sval[q];
    if (p >= (uint32_t)n) {
        nxt[i] = NONE;
        return;
    }
    if (q == 0 || skey[q - 1] != p)
        head[p] = (int32_t)i;
    nxt[i] = (q + 1 < n && skey[q + 1] == p) ? (int32_t)sval[q + 1] : NONE;
}

struct GatherArgs {
    const uint32_t *tgt;
    const int32_t *head, *nxt;
    int64_t n;
    int kind, nr;
    const int32_t *edge_user, *items, *neg;
    int32_t *u, *i;
    void *third;
    const DsStatus *st;
};

// perm[k] by the chase, then row perm[k] of the unshuffled sample (sampler.py:38,69,97)
__global__ __launch_bounds__(256) void perm_gather(GatherArgs a)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= a.n || a.st->status != DS_OK)
        return;
    int64_t s;
    const uint32_t t = k == 0 ? 0u : a.tgt[k];
    int32_t xi = (k == 0 || t == (uint32_t)k) ? (int32_t)k : a.nxt[k];
    if (xi == NONE) {
        s = t; // nobody touched the target before step k: it still holds its own index
    } else {
        for (int64_t hops = 0; hops < a.n; ++hops) { // the pointers lead to ever larger steps: the bound is never reached
            const int32_t h = a.head[xi];
            if (h == NONE)
                break;
            xi = h;
        }
        s = xi;
    }
    const int nr = a.nr;
    if (a.kind == 0) { // pairwise: row s = (positive s / nr, its (s % nr)-th negative)
        const int64_t e = s / nr;
        a.u[k] = a.edge_user[e];
        a.i[k] = a.items[e];
        ((int32_t *)a.third)[k] = a.neg[s];
    } else if (a.kind == 1) { // pointwise: row s = positive e (t == 0, label 1) or its (t-1)-th negative (label 0)
        const int64_t w = 1 + nr, e = s / w, tt = s - e * w;
        a.u[k] = a.edge_user[e];
        a.i[k] = tt == 0 ? a.items[e] : a.neg[e * nr + (tt - 1)];
        ((float *)a.third)[k] = tt == 0 ? 1.0f : 0.0f;
    } else { // cml: row s = positive s with its nr negatives
        a.u[k] = a.edge_user[s];
        a.i[k] = a.items[s];
        int32_t *row = (int32_t *)a.third + k * nr;
        for (int q = 0; q < nr; ++q)
            row[q] = a.neg[s * nr + q];
    }
}

} // namespace

// ---- the object behind the C ABI ---------------------------------------------------------------------------------------
struct crec_dsampler {
    int device = -1;
    int64_t n_users = 0, n_pos = 0;
    int32_t item_nums = 0;
    int bw = 0, nbuf = 0, pwords = 1;
    int64_t max_uniq = 0;
    double s1 = 0; // sum over users of deg / (item_nums - distinct items): expected candidates per unit of neg_ratio
    crec_devbuf ptr, items, edge_user, bitmaps, polys;
    crec_devbuf x, base, next_key, cand, cand_raw, counts, neg, tgt, key, val, skey, sval, head, nxt, sort_tmp, status;
    DsStatus *status_host = nullptr; // page-locked
    hipEvent_t done = nullptr;
    bool pending = false;
    // the call in flight (for the retry with larger budgets)
    int kind = 0, nr = 0;
    uint32_t key_in[MT_N];
    int32_t pos_in = 0;
    int32_t *u = nullptr, *i = nullptr;
    void *third = nullptr;
    hipStream_t stream = nullptr;
    double growth = 1.0;
    int64_t n = 0, raw_next1_in = 0;
    // the parallel form of the negatives phase: per-user degrees (host), the plan for one neg_ratio, its tables
    std::vector<int32_t> deg, uniq;
    int plan_nr = -1;      // the neg_ratio the plan below was made for (-1: none)
    bool plan_ok = false;  // the plan fits the kernels' limits
    bool force_walk = false, force_serial_shuffle = false;
    int64_t walk_fallbacks = 0;
    int32_t last_why = 0, last_why_at = 0;
    uint32_t plan_rcap = 0, plan_wcap = 0, plan_nseg = 0, plan_wfirst_max = 0, plan_rcap_a = 0, plan_wcap_a = 0, plan_na = 0, plan_nb = 0;
    crec_devbuf p_lo, p_W, p_len, p_toff, p_need, p_base, p_foff, p_ids, tabT, tabF, p_start, pr_cls, pr_blk;
};

namespace {

double harmonic_span(int64_t lo, int64_t hi) // sum_{i=lo}^{hi} 1 / (i + 1)
{
    if (hi < lo)
        return 0;
    if (hi - lo < 64) {
        double s = 0;
        for (int64_t i = lo; i <= hi; ++i)
            s += 1.0 / (double)(i + 1);
        return s;
    }
    return std::log(((double)hi + 1.5) / ((double)lo + 0.5));
}

// expected words of np.random.permutation(n): per step i one draw accepted with probability (i + 1) / (mask(i) + 1)
double expected_shuffle_words(int64_t n)
{
    double s = 0;
    int64_t hi = n - 1;
    while (hi >= 1) {
        const uint32_t m = mask_of((uint32_t)hi);
        const int64_t lo = (int64_t)(m >> 1) + 1;
        s += ((double)m + 1.0) * harmonic_span(lo, hi);
        hi = lo - 1;
    }
    return s;
}

int free_bufs(crec_dsampler *s)
{
    crec_devbuf *bufs[] = {&s->ptr, &s->items, &s->edge_user, &s->bitmaps, &s->polys, &s->x, &s->base, &s->next_key, &s->cand, &s->cand_raw,
                           &s->counts, &s->neg, &s->tgt, &s->key, &s->val, &s->skey, &s->sval, &s->head, &s->nxt, &s->sort_tmp, &s->status,
                           &s->p_lo, &s->p_W, &s->p_len, &s->p_toff, &s->p_need, &s->p_base, &s->p_foff, &s->p_ids, &s->tabT, &s->tabF, &s->p_start,
                           &s->pr_cls, &s->pr_blk};
    for (crec_devbuf *b : bufs)
        if (b->ptr) {
            (void)hipFree(b->ptr);
            b->ptr = nullptr;
            b->bytes = 0;
        }
    if (s->status_host)
        (void)hipHostFree(s->status_host);
    if (s->done)
        (void)hipEventDestroy(s->done);
    return 0;
}

// The plan of the parallel negatives phase for one neg_ratio: where every user's start is expected (mean and variance of the
// candidates the users before it take: need / (1 - p) and need p / (1 - p)^2 for a user that has seen the share p of the
// catalogue, plus the rare repeats), the window of starts its table covers (+- six standard deviations + 64) and how far its
// tested range reaches.  Host arithmetic in double, uploaded once per (sampler, neg_ratio).
int make_plan(crec_ctx *ctx, crec_dsampler *s, int nr, hipStream_t st)
{
    if (s->plan_nr == nr)
        return CREC_OK;
    s->plan_nr = nr;
    s->plan_ok = false;
    const int64_t U = s->n_users;
    if (nr < 1 || nr > TAB_NR_MAX || s->bw > WALK_BW_MAX || U < 1 || U > (int64_t)TAB_MAXSEG * TAB_SEG)
        return CREC_OK;
    std::vector<uint32_t> lo, W, len, toff, need, base, foff;
    try {
        lo.resize(U), W.resize(U), len.resize(U), toff.resize(U), need.resize(U), base.resize(U);
        foff.resize(crec_ceil_div(U, TAB_SEG));
    } catch (...) {
        return crec_fail(ctx, CREC_ENOMEM, "host allocation failed");
    }
    const double I = (double)s->item_nums;
    double S = 0, V = 0;
    uint64_t tsum = 0, fsum = 0, pos_base = 0;
    uint32_t rcap = 0, wcap = 0, wfirst = 0;
    for (int64_t k = 0; k < U; ++k) {
        const double nd = (double)s->deg[k] * nr, unseen = I - (double)s->uniq[k], p = (double)s->uniq[k] / I;
        const double mu = nd / (1.0 - p) * (1.0 + 0.5 * (nr - 1) / unseen), var = nd * p / ((1.0 - p) * (1.0 - p)) + nd * 0.5 * (nr - 1) / unseen;
        const double m = 6.0 * std::sqrt(V) + 64.0;
        const double a = std::floor(S - m), b = std::ceil(S + m);
        const double l = a < 0 ? 0 : a;
        const double span = b - l + 1.0 + std::ceil(mu + 6.0 * std::sqrt(var) + 64.0);
        if (b + span > 4.0e9)
            return CREC_OK; // beyond 32-bit positions: the walk
        lo[k] = (uint32_t)l, W[k] = (uint32_t)(b - l + 1.0), len[k] = (uint32_t)span;
        need[k] = (uint32_t)(s->deg[k] * (int64_t)nr), base[k] = (uint32_t)pos_base;
        pos_base += need[k];
        toff[k] = (uint32_t)tsum;
        tsum += W[k];
        if (k % TAB_SEG == 0) {
            foff[k / TAB_SEG] = (uint32_t)fsum;
            fsum += W[k];
            wfirst = W[k] > wfirst ? W[k] : wfirst;
        }
        rcap = len[k] > rcap ? len[k] : rcap;
        wcap = W[k] > wcap ? W[k] : wcap;
        S += mu, V += var;
    }
    const size_t lds = ((size_t)s->bw + rcap + TAB_EV + TAB_SK + 8) * 4 + (size_t)wcap * 2 + 4;
    if (tsum >= ((uint64_t)1 << 31) || lds > 150 * 1024 || rcap > 65535 || s->item_nums > 65536)
        return CREC_OK;
    // class A: ranges of at most 12288 candidates (three workgroups to a CU); class B: the rest
    std::vector<uint32_t> ids((size_t)U);
    uint32_t na = 0, nb = 0, rcap_a = 0, wcap_a = 0;
    for (int64_t k = 0; k < U; ++k)
        if (len[k] <= 12288)
            ids[na++] = (uint32_t)k, rcap_a = len[k] > rcap_a ? len[k] : rcap_a, wcap_a = W[k] > wcap_a ? W[k] : wcap_a;
    for (int64_t k = 0; k < U; ++k)
        if (len[k] > 12288)
            ids[na + nb++] = (uint32_t)k;
    const auto up = [&](crec_devbuf *b, const std::vector<uint32_t> &v) {
        if (int rc = crec_reserve(ctx, b, v.size() * 4 + 4))
            return rc;
        CREC_HIP(ctx, hipMemcpyAsync(b->ptr, v.data(), v.size() * 4, hipMemcpyHostToDevice, st));
        return (int)CREC_OK;
    };
    for (auto pr : {std::make_pair(&s->p_lo, &lo), std::make_pair(&s->p_W, &W), std::make_pair(&s->p_len, &len), std::make_pair(&s->p_toff, &toff),
                    std::make_pair(&s->p_need, &need), std::make_pair(&s->p_base, &base), std::make_pair(&s->p_foff, &foff), std::make_pair(&s->p_ids, &ids)})
        if (int rc = up(pr.first, *pr.second))
            return rc;
    if (int rc = crec_reserve(ctx, &s->tabT, (size_t)tsum * 4 + 4))
        return rc;
    if (int rc = crec_reserve(ctx, &s->tabF, (size_t)fsum * 4 + 4))
        return rc;
    if (int rc = crec_reserve(ctx, &s->p_start, (size_t)(U + 1) * 4))
        return rc;
    CREC_HIP(ctx, hipStreamSynchronize(st)); // the vectors go away
    s->plan_rcap = rcap, s->plan_wcap = wcap, s->plan_nseg = (uint32_t)foff.size(), s->plan_wfirst_max = wfirst;
    s->plan_rcap_a = rcap_a, s->plan_wcap_a = wcap_a, s->plan_na = na, s->plan_nb = nb;
    s->plan_ok = true;
    return CREC_OK;
}

// everything of one sample, enqueued on s->stream
int enqueue(crec_ctx *ctx, crec_dsampler *s)
{
    const crec_range sample_range("crec_dsampler_sample");
    hipStream_t st = s->stream;
    const int nr = s->nr;
    const int64_t n_pos = s->n_pos, n = s->n, n_neg = n_pos * nr;
    const uint32_t mx = (uint32_t)(s->item_nums - 1), mask = mask_of(mx);
    // budgets: expectation + slack, grown on a retry
    const double e1 = nr > 0 ? (double)nr * ((double)mask + 1.0) * s->s1 : 0.0;
    int64_t w1 = nr > 0 ? (int64_t)((e1 * 1.02 + 8.0 * std::sqrt(e1 + 1.0) + 4096.0) * s->growth) : 0;
    w1 = crec_ceil_div(w1, CAND_BLK) * CAND_BLK;
    const double e2 = expected_shuffle_words(n);
    const int64_t w2 = (int64_t)((e2 * 1.02 + 8.0 * std::sqrt(e2 + 1.0) + 4096.0) * s->growth);
    const int64_t x_blocks = crec_ceil_div((int64_t)s->pos_in + w1 + w2, MT_N) + 1;
    const int64_t x_words = x_blocks * MT_N;
    CREC_REQUIRE(ctx, x_words < ((int64_t)1 << 32) && w1 < ((int64_t)1 << 32), "epoch needs more than 2^32 stream words");
    const int64_t cand_blocks = w1 / CAND_BLK;
    if (int rc = crec_reserve(ctx, &s->x, (size_t)x_words * 4))
        return rc;
    if (int rc = crec_reserve(ctx, &s->cand, (size_t)(w1 + 1) * 4))
        return rc;
    if (int rc = crec_reserve(ctx, &s->cand_raw, (size_t)(w1 + 1) * 4))
        return rc;
    if (int rc = crec_reserve(ctx, &s->counts, (size_t)(cand_blocks + 1) * 4))
        return rc;
    if (int rc = crec_reserve(ctx, &s->neg, (size_t)(n_neg + 1) * 4))
        return rc;
    for (crec_devbuf *b : {&s->tgt, &s->key, &s->val, &s->skey, &s->sval, &s->nxt})
        if (int rc = crec_reserve(ctx, b, (size_t)(n + 1) * 4))
            return rc;
    if (int rc = crec_reserve(ctx, &s->head, (size_t)(n + 2) * 4))
        return rc;
    unsigned sort_bits = 1;
    while (sort_bits < 32 && ((uint64_t)1 << sort_bits) <= (uint64_t)n)
        ++sort_bits;
    size_t tmp_bytes = 0;
    if (n > 1) {
        CREC_HIP(ctx, rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const uint32_t *)s->key.ptr, (uint32_t *)s->skey.ptr, (const uint32_t *)s->val.ptr,
                                                (uint32_t *)s->sval.ptr, (size_t)n, 0u, sort_bits, st));
        if (int rc = crec_reserve(ctx, &s->sort_tmp, tmp_bytes + 16))
            return rc;
    }
    DsStatus *dst = (DsStatus *)s->status.ptr;
    CREC_HIP(ctx, hipMemsetAsync(dst, 0, sizeof(DsStatus), st));
    uint32_t *x = (uint32_t *)s->x.ptr;

    // phase 0: the raw stream, super-block by super-block
    CREC_HIP(ctx, hipMemcpyAsync(s->next_key.ptr, s->key_in, sizeof(s->key_in), hipMemcpyHostToDevice, st));
    constexpr int64_t SB_BLOCKS = (int64_t)crec_mtj::T * crec_mtj::LB;
    const size_t chain_lds = (size_t)(crec_mtj::DEG + MT_N + 2 * MT_N) * 4 + (size_t)crec_mtj::DEG * 2 + 64;
    CREC_HIP(ctx, hipFuncSetAttribute((const void *)mt_chains, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chain_lds));
    for (int64_t b0 = 0; b0 < x_blocks; b0 += SB_BLOCKS) {
        const int64_t blocks = (x_blocks - b0) < SB_BLOCKS ? (x_blocks - b0) : SB_BLOCKS;
        const int n_chains = (int)crec_ceil_div(blocks, crec_mtj::LB);
        const bool more = b0 + SB_BLOCKS < x_blocks;
        hipLaunchKernelGGL(mt_base, dim3(1), dim3(256), 0, st, (const uint32_t *)s->next_key.ptr, (uint32_t *)s->base.ptr, crec_mtj::BASE_BLOCKS);
        hipLaunchKernelGGL(mt_chains, dim3(n_chains + (more ? 1 : 0)), dim3(CHAIN_THREADS), chain_lds, st, (const uint32_t *)s->base.ptr,
                           (const uint32_t *)s->polys.ptr, x + (size_t)b0 * MT_N, blocks, n_chains, more ? (uint32_t *)s->next_key.ptr : nullptr);
    }
    CREC_HIP(ctx, hipGetLastError());

    // phase 1
    const int64_t r0 = s->pos_in;
    if (nr > 0 && n_pos > 0) {
        hipLaunchKernelGGL(cand_count, dim3((unsigned)cand_blocks), dim3(256), 0, st, (const uint32_t *)x, r0, w1, mask, mx, (uint32_t *)s->counts.ptr);
        hipLaunchKernelGGL(cand_scan, dim3(1), dim3(1024), 0, st, (uint32_t *)s->counts.ptr, (int)cand_blocks, dst);
        hipLaunchKernelGGL(cand_scatter, dim3((unsigned)cand_blocks), dim3(256), 0, st, (const uint32_t *)x, r0, w1, mask, mx,
                           (const uint32_t *)s->counts.ptr, (int32_t *)s->cand.ptr, (uint32_t *)s->cand_raw.ptr);
        NegArgs na{(const int32_t *)s->cand.ptr, (const uint32_t *)s->cand_raw.ptr, (const int64_t *)s->ptr.ptr, s->n_users,
                   (const uint32_t *)s->bitmaps.ptr, s->bw, s->nbuf, nr, (int32_t *)s->neg.ptr, dst, r0, 0u, (uint32_t)n_neg};
        na.cand_cap = (uint32_t)(s->cand.bytes / 4) & ~3u;
        if (int rc = make_plan(ctx, s, nr, st))
            return rc;
        if (s->plan_ok && !s->force_walk && ctx->knob_dsampler_general == 0) {
            // the parallel form: tables of every user over its window of possible starts, then look-ups
            TabArgs ta{(const int32_t *)s->cand.ptr, (const uint32_t *)s->bitmaps.ptr, s->bw, nr, (uint32_t)s->n_users,
                       (const uint32_t *)s->p_lo.ptr, (const uint32_t *)s->p_W.ptr, (const uint32_t *)s->p_len.ptr, (const uint32_t *)s->p_toff.ptr,
                       (const uint32_t *)s->p_need.ptr, (const uint32_t *)s->p_base.ptr, (uint32_t *)s->tabT.ptr, (uint32_t *)s->p_start.ptr, dst,
                       s->plan_rcap, s->plan_wcap, (int32_t *)s->neg.ptr, (const uint32_t *)s->cand_raw.ptr, r0};
            // two classes of users by the length of their tested ranges: the short ones several workgroups to a CU
            const auto tab_lds = [&](uint32_t rcap, uint32_t wcap) { return ((size_t)s->bw + rcap + TAB_EV + TAB_SK + 8) * 4 + (size_t)((wcap + 1) & ~1u) * 2; };
            const size_t lds_a = tab_lds(s->plan_rcap_a, s->plan_wcap_a), lds_b = tab_lds(s->plan_rcap, s->plan_wcap);
            CREC_HIP(ctx, hipFuncSetAttribute((const void *)neg_tab, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_a > lds_b ? lds_a : lds_b)));
            if (s->plan_na)
                hipLaunchKernelGGL(neg_tab, dim3(s->plan_na), dim3(256), lds_a, st, ta, (const uint32_t *)s->p_ids.ptr, s->plan_rcap_a, s->plan_wcap_a);
            if (s->plan_nb)
                hipLaunchKernelGGL(neg_tab, dim3(s->plan_nb), dim3(256), lds_b, st, ta, (const uint32_t *)s->p_ids.ptr + s->plan_na, s->plan_rcap, s->plan_wcap);
            hipLaunchKernelGGL(neg_compose, dim3(s->plan_nseg, (unsigned)crec_ceil_div(s->plan_wfirst_max, 256)), dim3(256), 0, st, ta,
                               (const uint32_t *)s->p_foff.ptr, (uint32_t *)s->tabF.ptr);
            hipLaunchKernelGGL(neg_chain_starts, dim3(1), dim3(64), 0, st, ta, (const uint32_t *)s->p_foff.ptr, (const uint32_t *)s->tabF.ptr, s->plan_nseg);
            // the users' true ranges are far shorter than the tested ones: most take the small launch
            const uint32_t emit_small = 4096;
            const auto emit_lds = [&](uint32_t rcap) { return ((size_t)s->bw + rcap + TAB_EV + TAB_SK + 8) * 4; };
            CREC_HIP(ctx, hipFuncSetAttribute((const void *)neg_emit, hipFuncAttributeMaxDynamicSharedMemorySize, (int)emit_lds(s->plan_rcap)));
            hipLaunchKernelGGL(neg_emit, dim3((unsigned)s->n_users), dim3(256), emit_lds(emit_small), st, ta, emit_small, 0u, emit_small);
            hipLaunchKernelGGL(neg_emit, dim3((unsigned)s->n_users), dim3(256), emit_lds(s->plan_rcap), st, ta, s->plan_rcap, emit_small + 1, 0xffffffffu);
        } else
        if (nr <= WALK_NR_MAX && s->bw <= WALK_BW_MAX && ctx->knob_dsampler_general != 1) {
            na.nbuf = WALK_NBUF;
            const size_t lds = ((size_t)WALK_NBUF * s->bw + CRING + (size_t)((nr - 1 + TILE + 1 + 3) & ~3) + 32 + 16 + UCACHE + 1) * 4;
            if (nr == 4) {
                CREC_HIP(ctx, hipFuncSetAttribute((const void *)neg_walk<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(neg_walk<4>, dim3(1), dim3(WT), lds, st, na);
            } else {
                CREC_HIP(ctx, hipFuncSetAttribute((const void *)neg_walk<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(neg_walk<0>, dim3(1), dim3(WT), lds, st, na);
            }
        } else {
            const size_t lds = ((((size_t)s->nbuf * s->bw + 1) & ~(size_t)1) + CBUF + ((nr - 1 + TILE + 1) & ~1)) * 4 + (size_t)(UCACHE + 1) * 8;
            if (s->pwords == 1) {
                CREC_HIP(ctx, hipFuncSetAttribute((const void *)neg_chain_general<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(neg_chain_general<1>, dim3(1), dim3(TILE), lds, st, na);
            } else {
                CREC_HIP(ctx, hipFuncSetAttribute((const void *)neg_chain_general<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(neg_chain_general<8>, dim3(1), dim3(TILE), lds, st, na);
            }
        }
    } else {
        // no draws in phase 1: phase 2 starts where the generator stands
        s->raw_next1_in = r0; // lives in the object: the copy is asynchronous
        CREC_HIP(ctx, hipMemcpyAsync(&dst->raw_next1, &s->raw_next1_in, sizeof(int64_t), hipMemcpyHostToDevice, st));
    }
    // phase 2: the regimes of large steps in parallel, the rest (and the final generator state) by the walk
    int64_t i0 = n - 1;
    int after_regimes = 0;
    if (!s->force_serial_shuffle && ctx->knob_dsampler_general == 0) {
        std::vector<PermRegime> regs;
        int64_t hi = n - 1;
        int nbmax = 0;
        while (hi >= PR_MIN_STEP) {
            const uint32_t m = mask_of((uint32_t)hi);
            const int64_t lo = (int64_t)(m >> 1) + 1, R = hi - lo + 1;
            // accepted with probability (i + 1) / (m + 1) >= 1/2: 2 R words on average at worst, + eight standard deviations
            const int64_t tmax = 2 * R + (int64_t)(8.0 * std::sqrt(2.0 * (double)R)) + 512;
            PermRegime g{(int32_t)hi, (int32_t)R, m, (uint32_t)tmax, (int)crec_ceil_div(tmax, PR_BLK), regs.empty() ? 1 : 0};
            nbmax = g.nb > nbmax ? g.nb : nbmax;
            regs.push_back(g);
            hi = lo - 1;
        }
        if (!regs.empty()) {
            if (int rc = crec_reserve(ctx, &s->pr_cls, (size_t)nbmax * PR_BLK + 64))
                return rc;
            if (int rc = crec_reserve(ctx, &s->pr_blk, (size_t)(PR_ROUNDS * 2 + 1) * nbmax * 4 + 64))
                return rc;
            uint32_t *blk = (uint32_t *)s->pr_blk.ptr, *fin = blk + (size_t)PR_ROUNDS * 2 * nbmax;
            for (const PermRegime &g : regs) {
                for (int r = 0; r < PR_ROUNDS; ++r)
                    hipLaunchKernelGGL(perm_round, dim3(g.nb), dim3(256), 0, st, (const uint32_t *)x, (uint32_t)x_words, g, r, (uint8_t *)s->pr_cls.ptr, blk,
                                       nbmax, (const DsStatus *)dst);
                hipLaunchKernelGGL(perm_count, dim3(g.nb), dim3(256), 0, st, g, (const uint8_t *)s->pr_cls.ptr, fin, dst);
                hipLaunchKernelGGL(perm_emit, dim3(g.nb), dim3(256), 0, st, (const uint32_t *)x, g, (const uint8_t *)s->pr_cls.ptr, (const uint32_t *)fin,
                                   (uint32_t *)s->tgt.ptr, dst);
            }
            i0 = hi;
            after_regimes = 1;
        }
    }
    hipLaunchKernelGGL(perm_targets, dim3(1), dim3(WT), 0, st, (const uint32_t *)x, x_words, i0, after_regimes, (uint32_t *)s->tgt.ptr, dst);
    // phase 3
    if (n > 0) {
        const unsigned g = (unsigned)crec_ceil_div(n, 256);
        hipLaunchKernelGGL(perm_keys, dim3(g), dim3(256), 0, st, (const uint32_t *)s->tgt.ptr, n, (uint32_t *)s->key.ptr, (uint32_t *)s->val.ptr,
                           (int32_t *)s->head.ptr, (const DsStatus *)dst);
        const uint32_t *skey = (const uint32_t *)s->key.ptr, *sval = (const uint32_t *)s->val.ptr;
        if (n > 1) {
            CREC_HIP(ctx, rocprim::radix_sort_pairs(s->sort_tmp.ptr, tmp_bytes, (const uint32_t *)s->key.ptr, (uint32_t *)s->skey.ptr,
                                                    (const uint32_t *)s->val.ptr, (uint32_t *)s->sval.ptr, (size_t)n, 0u, sort_bits, st));
            skey = (const uint32_t *)s->skey.ptr;
            sval = (const uint32_t *)s->sval.ptr;
        }
        hipLaunchKernelGGL(perm_links, dim3(g), dim3(256), 0, st, skey, sval, n, (int32_t *)s->head.ptr, (int32_t *)s->nxt.ptr, (const DsStatus *)dst);
        GatherArgs ga{(const uint32_t *)s->tgt.ptr, (const int32_t *)s->head.ptr, (const int32_t *)s->nxt.ptr, n, s->kind, nr,
                      (const int32_t *)s->edge_user.ptr, (const int32_t *)s->items.ptr, (const int32_t *)s->neg.ptr, s->u, s->i, s->third,
                      (const DsStatus *)dst};
        hipLaunchKernelGGL(perm_gather, dim3(g), dim3(256), 0, st, ga);
    }
    CREC_HIP(ctx, hipGetLastError());
    CREC_HIP(ctx, hipMemcpyAsync(s->status_host, dst, sizeof(DsStatus), hipMemcpyDeviceToHost, st));
    CREC_HIP(ctx, hipEventRecord(s->done, st));
    s->pending = true;
    return CREC_OK;
}

} // namespace

extern "C" {

int crec_dsampler_create(crec_ctx *ctx, const int64_t *ptr, const int32_t *items, int64_t n_users, const int32_t *user_ids, int32_t item_nums,
                         crec_dsampler **out, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, ptr && user_ids && out && n_users >= 0 && item_nums >= 1, "null argument / bad sizes");
    *out = nullptr;
    const int64_t n_pos = ptr[n_users];
    CREC_REQUIRE(ctx, items || n_pos == 0, "null items");
    for (int64_t k = 0; k < n_users; ++k)
        CREC_REQUIRE(ctx, ptr[k + 1] >= ptr[k], "ptr not monotone at %lld", (long long)k);
    // what the kernels are built for; anything else stays with the host sampler (crec_sample_*)
    const int bw = (int)(crec_ceil_div(crec_ceil_div(item_nums, 32), 256) * 256); // words per bitmap row: whole 1 KiB LDS-DMA pieces
    if (bw > 8 * TILE || (double)n_users * bw * 4.0 > 2147483648.0)
        return crec_fail(ctx, CREC_EUNSUPPORTED, "device sampler: item_nums <= 262144 and users x items / 8 <= 2 GiB (use crec_sample_*)");
    std::vector<int32_t> stamp, eu;
    try {
        stamp.assign((size_t)item_nums, 0);
        eu.resize((size_t)(n_pos > 0 ? n_pos : 1));
    } catch (...) {
        return crec_fail(ctx, CREC_ENOMEM, "host allocation failed");
    }
    double s1 = 0;
    int64_t max_uniq = 0;
    std::vector<int32_t> degs, uniqs;
    try {
        degs.resize((size_t)n_users), uniqs.resize((size_t)n_users);
    } catch (...) {
        return crec_fail(ctx, CREC_ENOMEM, "host allocation failed");
    }
    for (int64_t k = 0; k < n_users; ++k) {
        int64_t uniq = 0;
        for (int64_t e = ptr[k]; e < ptr[k + 1]; ++e) {
            CREC_REQUIRE(ctx, items[e] >= 0 && items[e] < item_nums, "item id %d out of range", items[e]);
            uniq += stamp[items[e]] != (int32_t)(k + 1);
            stamp[items[e]] = (int32_t)(k + 1);
            eu[(size_t)e] = user_ids[k];
        }
        max_uniq = uniq > max_uniq ? uniq : max_uniq;
        degs[(size_t)k] = (int32_t)(ptr[k + 1] - ptr[k]), uniqs[(size_t)k] = (int32_t)uniq;
        if (uniq < item_nums)
            s1 += (double)(ptr[k + 1] - ptr[k]) / (double)(item_nums - uniq);
    }
    const char *jerr = nullptr;
    const uint32_t *tab = crec_mtj::jump_table(&jerr);
    if (!tab)
        return crec_fail(ctx, CREC_EINVAL, "%s", jerr ? jerr : "MT19937 jump table unavailable");
    crec_dsampler *s = new (std::nothrow) crec_dsampler();
    if (!s)
        return crec_fail(ctx, CREC_ENOMEM, "host allocation failed");
    s->device = ctx->device;
    s->n_users = n_users, s->n_pos = n_pos, s->item_nums = item_nums, s->bw = bw, s->max_uniq = max_uniq, s->s1 = s1;
    s->pwords = bw <= TILE ? 1 : 8;
    s->deg.swap(degs), s->uniq.swap(uniqs);
    {
        const int fit = (int)((96 * 1024) / ((size_t)bw * 4)); // ring of bitmaps: as many users ahead as 96 KB hold, 3 .. 6
        s->nbuf = fit < 3 ? 3 : (fit > 6 ? 6 : fit);
    }
    hipStream_t st = (hipStream_t)stream;
    const auto fail = [&](int rc) {
        free_bufs(s);
        delete s;
        return rc;
    };
#define DS_TRY(expr)               \
    do {                           \
        if (int rc_ = (expr))      \
            return fail(rc_);      \
    } while (0)
#define DS_HIP(call)                                                                                        \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(crec_fail(ctx, CREC_EHIP, "%s failed: %s", #call, hipGetErrorString(e_)));          \
    } while (0)
    DS_HIP(hipSetDevice(ctx->device));
    DS_TRY(crec_reserve(ctx, &s->ptr, (size_t)(n_users + 1) * 8));
    DS_TRY(crec_reserve(ctx, &s->items, (size_t)(n_pos + 1) * 4));
    DS_TRY(crec_reserve(ctx, &s->edge_user, (size_t)(n_pos + 1) * 4));
    DS_TRY(crec_reserve(ctx, &s->bitmaps, (size_t)(n_users + 1) * bw * 4));
    DS_TRY(crec_reserve(ctx, &s->polys, (size_t)crec_mtj::T * crec_mtj::POLY_WORDS * 4));
    DS_TRY(crec_reserve(ctx, &s->base, (size_t)crec_mtj::BASE_BLOCKS * MT_N * 4));
    DS_TRY(crec_reserve(ctx, &s->next_key, (size_t)MT_N * 4));
    DS_TRY(crec_reserve(ctx, &s->status, sizeof(DsStatus)));
    DS_HIP(hipHostMalloc((void **)&s->status_host, sizeof(DsStatus), hipHostMallocDefault));
    DS_HIP(hipEventCreateWithFlags(&s->done, hipEventDisableTiming));
    DS_HIP(hipMemcpyAsync(s->ptr.ptr, ptr, (size_t)(n_users + 1) * 8, hipMemcpyHostToDevice, st));
    if (n_pos > 0) {
        DS_HIP(hipMemcpyAsync(s->items.ptr, items, (size_t)n_pos * 4, hipMemcpyHostToDevice, st));
        DS_HIP(hipMemcpyAsync(s->edge_user.ptr, eu.data(), (size_t)n_pos * 4, hipMemcpyHostToDevice, st));
    }
    DS_HIP(hipMemcpyAsync(s->polys.ptr, tab, (size_t)crec_mtj::T * crec_mtj::POLY_WORDS * 4, hipMemcpyHostToDevice, st));
    DS_HIP(hipMemsetAsync(s->bitmaps.ptr, 0, (size_t)(n_users + 1) * bw * 4, st));
    if (n_users > 0 && n_pos > 0)
        hipLaunchKernelGGL(build_bitmaps, dim3((unsigned)n_users), dim3(256), 0, st, (const int64_t *)s->ptr.ptr, (const int32_t *)s->items.ptr, bw,
                           (uint32_t *)s->bitmaps.ptr);
    DS_HIP(hipGetLastError());
    DS_HIP(hipStreamSynchronize(st)); // eu / the caller's arrays may go away
#undef DS_TRY
#undef DS_HIP
    *out = s;
    return CREC_OK;
}

void crec_dsampler_destroy(crec_dsampler *s)
{
    if (!s)
        return;
    if (s->device >= 0) {
        (void)hipSetDevice(s->device);
        if (s->pending && s->done)
            (void)hipEventSynchronize(s->done);
    }
    free_bufs(s);
    delete s;
}

int crec_dsampler_sample(crec_ctx *ctx, crec_dsampler *s, int kind, int neg_ratio, const uint32_t key[624], int32_t pos, int32_t *u, int32_t *i,
                         void *third, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, s && key && pos >= 0 && pos <= MT_N, "null sampler / bad MT19937 state");
    CREC_REQUIRE(ctx, kind >= 0 && kind <= 2 && neg_ratio >= 0 && neg_ratio <= 1024, "kind 0..2, 0 <= neg_ratio <= 1024");
    CREC_REQUIRE(ctx, !s->pending, "the previous sample has not been finished (crec_dsampler_finish)");
    const int64_t n = kind == 0 ? s->n_pos * neg_ratio : (kind == 1 ? s->n_pos * (1 + (int64_t)neg_ratio) : s->n_pos);
    CREC_REQUIRE(ctx, n == 0 || (u && i && third), "null output");
    CREC_REQUIRE(ctx, n < ((int64_t)1 << 31) && s->n_pos * (int64_t)neg_ratio < ((int64_t)1 << 31), "epoch too large for the device sampler");
    // the reference's rejection loop ends only if every user can supply neg_ratio unseen items (crec_sample_* checks the same)
    CREC_REQUIRE(ctx, s->n_pos == 0 || neg_ratio == 0 || s->max_uniq + neg_ratio <= s->item_nums, "a user has fewer than neg_ratio unseen items");
    CREC_HIP(ctx, hipSetDevice(s->device));
    s->kind = kind, s->nr = neg_ratio, s->pos_in = pos, s->u = u, s->i = i, s->third = third, s->stream = (hipStream_t)stream;
    s->n = n;
    s->force_walk = s->force_serial_shuffle = false;
    s->growth = ctx->knob_dsampler_budget > 0 ? ctx->knob_dsampler_budget / 100.0 : 1.0;
    std::memcpy(s->key_in, key, sizeof(s->key_in));
    return enqueue(ctx, s);
}

int crec_dsampler_finish(crec_ctx *ctx, crec_dsampler *s, int64_t *n_out, uint32_t key_out[624], int32_t *pos_out)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, s && n_out && key_out && pos_out, "null argument");
    CREC_REQUIRE(ctx, s->pending, "no sample in flight");
    CREC_HIP(ctx, hipSetDevice(s->device));
    for (int attempt = 0;; ++attempt) {
        CREC_HIP(ctx, hipEventSynchronize(s->done));
        s->pending = false;
        if (s->status_host->status == DS_OK)
            break;
        // the stream budget (sized from the expected rejection counts) did not last: again with twice the words
        CREC_REQUIRE(ctx, s->status_host->status != DS_INTERNAL, "device sampler: internal error (undecided words left in a shuffle tile)");
        if (s->status_host->status == DS_NEED_SERIAL_SHUFFLE && !s->force_serial_shuffle) { // a regime not settled within its rounds
            s->force_serial_shuffle = true;
            ++s->walk_fallbacks;
            s->last_why = 6, s->last_why_at = 0;
            if (int rc = enqueue(ctx, s))
                return rc;
            continue;
        }
        if (s->status_host->status == DS_NEED_WALK && !s->force_walk) { // the parallel negatives phase met something it reports instead of handling
            s->force_walk = true;
            ++s->walk_fallbacks;
            s->last_why = s->status_host->why, s->last_why_at = s->status_host->why_at;
            if (int rc = enqueue(ctx, s))
                return rc;
            continue;
        }
        CREC_REQUIRE(ctx, attempt < 6, "device sampler: stream budget exhausted after %d retries (status %d)", attempt, s->status_host->status);
        s->growth *= 2.0;
        if (int rc = enqueue(ctx, s))
            return rc;
    }
    *n_out = s->n;
    std::memcpy(key_out, s->status_host->key_out, sizeof(uint32_t) * MT_N);
    *pos_out = s->status_host->pos_out;
    return CREC_OK;
}

// what the last finished sample consumed: stream words of the negatives phase, of the shuffle, candidates produced / used
int crec_dsampler_stats(const crec_dsampler *s, int64_t out[13])
{
    if (!s || !out || !s->status_host)
        return CREC_EINVAL;
    out[0] = s->status_host->raw_next1 - s->pos_in;
    out[1] = s->status_host->raw_next2 - s->status_host->raw_next1;
    out[2] = s->status_host->n_cand;
    out[3] = s->status_host->cand_used;
    for (int k = 0; k < 6; ++k)
        out[4 + k] = s->status_host->dbg[k];
    out[10] = s->walk_fallbacks;
    out[11] = s->last_why, out[12] = s->last_why_at;
    return CREC_OK;
}

} // extern "C"
