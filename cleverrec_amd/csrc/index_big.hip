// index_big.hip: the lazy step's index for batches of ANY size (the reference takes any `batch_size`: model/Recommender.py:17).
//
// bpr_index_batches sorts a batch's (row, slot) pairs in LDS -- at most 16 384 pairs, batches of at most 8 192 triples.  Here the
// same index is built in global memory for a whole chunk of batches at once, every phase a flat launch over all of the chunk's
// elements (no workgroup owns a batch, so nothing grows with the batch):
//
//   big_fill         (row, slot) pairs of both sides of every batch; per batch a user segment and an item segment (i-slots, then
//                    j-slots, so that a stable sort leaves a row's occurrences in batch order, positives first)
//   sort_*           segmented stable radix sort by row over the significant bits of a row id, every segment cut into tiles
//   big_runs         per sorted element: its row's segment [start, start + cnt) (two binary searches, skipped where the
//                    neighbours already tell), on the item side the row's positive slots `ci` (a third); the user side's
//                    descriptors are complete here
//   rocPRIM scan     over the front flags (positive slots of rows with two or more of them): the processing slots of the
//                    item-sorted front region
//   big_item_desc    the item descriptors by batch position, the front region's order, every position's busier item row
//   big_part_flags + rocPRIM scan + big_emit_order   the stable partition behind the front region: positions with another
//                    multi-occurrence item row, then the rest
//   big_gather       {position, user row} and the two item descriptors by processing slot
//
// The result equals bpr_index_batches' lazy branch bit for bit (tests/test_gpu_bigbatch.py holds the two against each other on
// batches both can build); the triple pass does not know which builder ran.
#include "crec_internal.h"
#include "lazy_desc.h"
#include "shard_index.h"

#include <rocprim/rocprim.hpp>

namespace {

struct BigDims {
    int64_t n_chunk; // triples in the chunk
    int32_t B;       // triples per batch (the last batch of the chunk may be shorter)
    int32_t nbc;
    __device__ __forceinline__ int nb_of(int b) const
    {
        const int64_t left = n_chunk - (int64_t)b * B;
        return (int)(left < B ? left : B);
    }
};

// element e of the chunk's pair space: batch e / 3B; inside a batch [0, B) the user side, [B, 3B) the item side
struct BigAt {
    int b, side, k, n, nb;
    int64_t seg; // first element of the segment
};
__device__ __forceinline__ BigAt big_at(const BigDims &dm, int b, int r) // r: the element's place inside the batch's 3B
{
    BigAt a;
    const int64_t per = 3 * (int64_t)dm.B;
    a.b = b;
    a.nb = dm.nb_of(a.b);
    a.side = r >= dm.B ? 1 : 0;
    a.k = a.side ? r - dm.B : r;
    a.n = a.side ? 2 * a.nb : a.nb;
    a.seg = (int64_t)a.b * per + (a.side ? dm.B : 0);
    return a;
}

__global__ __launch_bounds__(256) void big_fill(const int32_t *__restrict__ u, const int32_t *__restrict__ i, const int32_t *__restrict__ j,
                                                BigDims dm, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                int64_t *__restrict__ seg0, int32_t *__restrict__ segn)
{
    if (blockIdx.x == 0 && threadIdx.x < 2) { // segment 2 * batch + side
        const int b = (int)blockIdx.y, side = (int)threadIdx.x, nb = dm.nb_of(b);
        seg0[2 * b + side] = 3 * (int64_t)dm.B * b + (side ? dm.B : 0);
        segn[2 * b + side] = side ? 2 * nb : nb;
    }
    // (the launches over a chunk's elements: grid.y = the batch, grid.x strides over the batch's elements -- no 64-bit divisions)
    const int b = (int)blockIdx.y;
    for (int r = (int)(blockIdx.x * blockDim.x + threadIdx.x); r < 3 * dm.B; r += (int)(gridDim.x * blockDim.x)) {
        const BigAt a = big_at(dm, b, r);
        const int64_t e = 3 * (int64_t)dm.B * b + r;
        uint32_t key = 0xffffffffu;
        if (a.k < a.n) {
            const int64_t t0 = (int64_t)a.b * dm.B;
            key = (uint32_t)(a.side == 0 ? u[t0 + a.k] : (a.k < a.nb ? i[t0 + a.k] : j[t0 + a.k - a.nb]));
        }
        keys[e] = key;
        vals[e] = (uint32_t)a.k;
    }
}

// ---- the sort: a segmented, stable least-significant-digit radix sort, 8 bits per pass, every segment cut into tiles of
// SORT_TILE elements so that a pass is three flat launches over ALL tiles of ALL segments (rocPRIM's segmented sort gives a
// long segment to ONE workgroup: 1.9 ms for a chunk of 64 batches of 65 536 whatever the chunk holds; this one 0.44 ms):
//   sort_hist     per tile: how many of its elements carry each digit            -> hist[segment][digit][tile]
//   sort_scan     per segment: exclusive scan in (digit, tile) order = where each tile's elements of a digit go
//   sort_scatter  per tile: every element's rank among the tile's elements of its digit, IN THE TILE'S ORDER (stable) -- a wave
//                 owns a contiguous 512 elements of the tile, the lanes holding the same digit find each other with one ballot per
//                 digit bit (block_sort.h's scheme), per-wave digit counts are summed in wave order -- and the element moves
#ifndef CREC_SORT_THREADS
#define CREC_SORT_THREADS 256 // tiles of 2048.  (1024 threads / tiles of 8192, for longer runs in the scatter: sort_scatter 89 -> 132 us, sort_hist 24 -> 37 us per pass over 12.6 M pairs)
#endif
constexpr int SORT_THREADS = CREC_SORT_THREADS, SORT_WAVES = SORT_THREADS / 64, SORT_PER_WAVE = 512, SORT_TILE = SORT_WAVES * SORT_PER_WAVE;

// the segments of one sort: segment s holds segn[s] pairs from element seg0[s] on; a launch has a workgroup per (tile, segment)
struct SegSort {
    const int64_t *seg0; // [nseg] device
    const int32_t *segn; // [nseg] device
    int tiles_max;       // tiles of the longest segment
};

__global__ __launch_bounds__(SORT_THREADS) void sort_hist(SegSort sd, const uint32_t *__restrict__ keys, int shift, int32_t *__restrict__ hist)
{
    __shared__ int cnt[256];
    const int seg = (int)blockIdx.y, tile = (int)blockIdx.x, n = sd.segn[seg];
    const uint32_t *sk = keys + sd.seg0[seg];
    if (threadIdx.x < 256)
        cnt[threadIdx.x] = 0;
    __syncthreads();
    const int k0 = tile * SORT_TILE;
    for (int k = k0 + (int)threadIdx.x; k < k0 + SORT_TILE && k < n; k += SORT_THREADS)
        atomicAdd(&cnt[(sk[k] >> shift) & 255u], 1);
    __syncthreads();
    if (threadIdx.x < 256)
        hist[((int64_t)seg * 256 + threadIdx.x) * sd.tiles_max + tile] = cnt[threadIdx.x]; // (a tile behind the segment's end: zeros)
}

__global__ __launch_bounds__(1024) void sort_scan(SegSort sd, int32_t *__restrict__ hist)
{
    __shared__ int wave_sums[1024 / 64 + 2];
    const int seg = (int)blockIdx.x;
    int32_t *h = hist + (int64_t)seg * 256 * sd.tiles_max;
    // the scan runs over (digit, tile), which is how the counters lie in memory; a thread owns a contiguous span of it
    const int total = 256 * sd.tiles_max, per = (total + 1023) / 1024;
    const int v0 = (int)threadIdx.x * per, v1 = (v0 + per) < total ? (v0 + per) : total;
    int sum = 0;
    for (int v = v0; v < v1; ++v)
        sum += h[v];
    // (an exclusive scan over the workgroup's threads, as block_sort.h's)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off)
            incl += t;
    }
    if (lane == 63)
        wave_sums[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        int sv = lane < 16 ? wave_sums[lane] : 0, si = sv;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const int t = __shfl_up(si, off, 64);
            if (lane >= off)
                si += t;
        }
        if (lane < 16)
            wave_sums[lane] = si - sv;
    }
    __syncthreads();
    int run = wave_sums[wave] + incl - sum;
    for (int v = v0; v < v1; ++v) {
        const int c = h[v];
        h[v] = run;
        run += c;
    }
}

__global__ __launch_bounds__(SORT_THREADS) void sort_scatter(SegSort sd, const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                             int shift, const int32_t *__restrict__ hist, uint32_t *__restrict__ keys_out,
                                                             uint32_t *__restrict__ vals_out)
{
    __shared__ int wcnt[SORT_WAVES][257]; // per wave: running count per digit
    __shared__ int base[256];             // per digit: where this tile's elements of the digit start in the segment
    const int seg = (int)blockIdx.y, tile = (int)blockIdx.x, n = sd.segn[seg];
    const int64_t seg0 = sd.seg0[seg];
    if (tile * SORT_TILE >= n)
        return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    constexpr int ROUNDS = SORT_PER_WAVE / 64;
    uint32_t r[ROUNDS], p[ROUNDS];
    int loc[ROUNDS];
    const int w0 = tile * SORT_TILE + wave * SORT_PER_WAVE;
#pragma unroll
    for (int s = 0; s < ROUNDS; ++s) {
        const int k = w0 + s * 64 + lane;
        r[s] = k < n ? keys[seg0 + k] : 0u;
        p[s] = k < n ? vals[seg0 + k] : 0u;
    }
    for (int c = lane; c < 256; c += 64)
        wcnt[wave][c] = 0;
    if (tid < 256)
        base[tid] = hist[((int64_t)seg * 256 + tid) * sd.tiles_max + tile];
    // rank inside the wave: LDS operations of one wave execute in order, so the running counter written in round s is the
    // one read in round s + 1
#pragma unroll
    for (int s = 0; s < ROUNDS; ++s) {
        const bool valid = w0 + s * 64 + lane < n;
        const uint32_t digit = (r[s] >> shift) & 255u;
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const bool on = (digit >> bit) & 1u;
            const uint64_t bal = __ballot(on);
            peers &= on ? bal : ~bal;
        }
        const int before = __popcll(peers & lt_mask), cnt = __popcll(peers);
        const int run = wcnt[wave][digit];
        loc[s] = run + before;
        if (valid && before == 0)
            wcnt[wave][digit] = run + cnt;
    }
    __syncthreads();
    if (tid < 256) { // per digit: the waves' counts -> their offsets, in wave order
        int run = base[tid];
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) {
            const int c = wcnt[w][tid];
            wcnt[w][tid] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < ROUNDS; ++s) {
        if (w0 + s * 64 + lane < n) {
            const int64_t dst = seg0 + wcnt[wave][(r[s] >> shift) & 255u] + loc[s];
            keys_out[dst] = r[s];
            vals_out[dst] = p[s];
        }
    }
}

// (ka, va) -> sorted by the low `bits` bits of the key, every segment by itself, equal keys in their incoming order.  The passes
// ping-pong between (ka, va) and (kb, vb); *kres / *vres: where the result is.  hist: 256 * tiles_max ints per segment.
inline void seg_sort(const SegSort &sd, int nseg, int bits, uint32_t *ka, uint32_t *va, uint32_t *kb, uint32_t *vb, int32_t *hist, hipStream_t st,
                     uint32_t **kres, uint32_t **vres)
{
    const dim3 tiles((unsigned)sd.tiles_max, (unsigned)nseg);
    uint32_t *ks = ka, *vs = va, *kd = kb, *vd = vb;
    for (int pass = 0; pass < (bits + 7) / 8; ++pass) {
        hipLaunchKernelGGL(sort_hist, tiles, dim3(SORT_THREADS), 0, st, sd, (const uint32_t *)ks, 8 * pass, hist);
        hipLaunchKernelGGL(sort_scan, dim3((unsigned)nseg), dim3(1024), 0, st, sd, hist);
        hipLaunchKernelGGL(sort_scatter, tiles, dim3(SORT_THREADS), 0, st, sd, (const uint32_t *)ks, (const uint32_t *)vs, 8 * pass,
                           (const int32_t *)hist, kd, vd);
        uint32_t *t = ks;
        ks = kd;
        kd = t;
        t = vs;
        vs = vd;
        vd = t;
    }
    *kres = ks;
    *vres = vs;
}
inline int sort_tiles(int64_t longest) { return (int)((longest + SORT_TILE - 1) / SORT_TILE > 0 ? (longest + SORT_TILE - 1) / SORT_TILE : 1); }

// first index in [lo, hi) of the sorted keys that is >= x (UPPER: > x)
template <bool UPPER> __device__ __forceinline__ int big_bound(const uint32_t *__restrict__ keys, int lo, int hi, uint32_t x)
{
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const uint32_t v = keys[mid];
        if (UPPER ? v <= x : v < x)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void big_runs(BigDims dm, const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                int32_t *__restrict__ rstart, int32_t *__restrict__ rcnt, int32_t *__restrict__ rci,
                                                int32_t *__restrict__ fp, int4 *__restrict__ u_desc)
{
    const int b = (int)blockIdx.y;
    if (b == 0 && blockIdx.x == 0 && threadIdx.x == 0)
        fp[3 * (int64_t)dm.B * dm.nbc] = 0; // the scan's closing element
    for (int r = (int)(blockIdx.x * blockDim.x + threadIdx.x); r < 3 * dm.B; r += (int)(gridDim.x * blockDim.x)) {
        const BigAt a = big_at(dm, b, r);
        const int64_t e = 3 * (int64_t)dm.B * b + r;
        int front_pos = 0;
        if (a.k < a.n) {
            const uint32_t *sk = keys + a.seg;
            const uint32_t row = sk[a.k];
            // most rows occur once: the neighbours tell, no search
            const int start = (a.k == 0 || sk[a.k - 1] != row) ? a.k : big_bound<false>(sk, 0, a.k, row);
            const int end = (a.k + 1 == a.n || sk[a.k + 1] != row) ? a.k + 1 : big_bound<true>(sk, a.k + 1, a.n, row);
            const int cnt = end - start;
            int ci = 0;
            if (a.side == 1 && cnt >= 2) { // the run's i-slots come first (stable sort): their number
                const uint32_t *sv = vals + a.seg;
                ci = big_bound<false>(sv, start, end, (uint32_t)a.nb) - start;
                ci = ci >= 2 ? ci : 0;
            }
            rstart[e] = start;
            rcnt[e] = cnt;
            rci[e] = ci;
            const int slot = (int)vals[e];
            if (a.side == 0)
                u_desc[(int64_t)a.b * dm.B + slot] = lazy_desc_pack(row, cnt == 1, a.k, start, cnt, 0, 0);
            else
                front_pos = (ci >= 2 && slot < a.nb) ? 1 : 0;
        }
        fp[e] = front_pos;
    }
}

__global__ __launch_bounds__(256) void big_item_desc(BigDims dm, int spw, const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                     const int32_t *__restrict__ rstart, const int32_t *__restrict__ rcnt,
                                                     const int32_t *__restrict__ rci, const int32_t *__restrict__ fg,
                                                     int4 *__restrict__ i_tmp, int4 *__restrict__ j_tmp, int32_t *__restrict__ order,
                                                     int32_t *__restrict__ wi, int32_t *__restrict__ wj)
{
    const int b = (int)blockIdx.y, nb = dm.nb_of(b);
    for (int k = (int)(blockIdx.x * blockDim.x + threadIdx.x); k < 2 * nb; k += (int)(gridDim.x * blockDim.x)) {
        const int64_t seg = 3 * (int64_t)dm.B * b + dm.B, e = seg + k;
        const int start = rstart[e], cnt = rcnt[e], ci = rci[e];
        const bool front = ci >= 2;
        const int first = front ? fg[seg + start] - fg[seg] : 0; // the row's first processing slot
        const int slot = (int)vals[e], pos = slot < nb ? slot : slot - nb;
        const int4 desc = lazy_desc_pack(keys[e], cnt == 1, k, start, cnt, ci, front ? first % spw : 0);
        const int64_t at = (int64_t)b * dm.B + pos;
        if (slot < nb) {
            i_tmp[at] = desc;
            if (front)
                order[(int64_t)b * dm.B + first + (k - start)] = slot;
            wi[at] = front ? -1 : cnt;
        } else {
            j_tmp[at] = desc;
            wj[at] = cnt;
        }
    }
}

// per batch position: {in the front region, has another multi-occurrence item row} as the two halves of one scan element
__global__ __launch_bounds__(256) void big_part_flags(BigDims dm, const int32_t *__restrict__ wi, const int32_t *__restrict__ wj,
                                                      unsigned long long *__restrict__ pf)
{
    const int b = (int)blockIdx.y, nb = dm.nb_of(b);
    if (b == 0 && blockIdx.x == 0 && threadIdx.x == 0)
        pf[(int64_t)dm.B * dm.nbc] = 0; // the scan's closing element
    for (int pos = (int)(blockIdx.x * blockDim.x + threadIdx.x); pos < dm.B; pos += (int)(gridDim.x * blockDim.x)) {
        const int64_t t = (int64_t)b * dm.B + pos;
        unsigned long long f = 0;
        if (pos < nb) {
            const int a = wi[t], c = wj[t];
            if (a == -1)
                f = 1ull << 32;
            else if ((a > c ? a : c) >= 2)
                f = 1ull;
        }
        pf[t] = f;
    }
}

__global__ __launch_bounds__(256) void big_emit_order(BigDims dm, const unsigned long long *__restrict__ pg, int32_t *__restrict__ order,
                                                      int32_t *__restrict__ nslots)
{
    const int b = (int)blockIdx.y, nb = dm.nb_of(b);
    for (int pos = (int)(blockIdx.x * blockDim.x + threadIdx.x); pos < nb; pos += (int)(gridDim.x * blockDim.x)) {
        const int64_t t0 = (int64_t)b * dm.B, t = t0 + pos;
        const unsigned long long base = pg[t0], tot = pg[t0 + nb] - base, before = pg[t] - base, mine = pg[t + 1] - pg[t];
        const int n_front = (int)(tot >> 32), tot_multi = (int)(tot & 0xffffffffull);
        const int bf = (int)(before >> 32), bm = (int)(before & 0xffffffffull);
        if (pos == 0) {
            nslots[b * 2 + 0] = nb;
            nslots[b * 2 + 1] = n_front;
        }
        if (mine >> 32)
            continue; // a front position: placed by big_item_desc
        if (mine & 1ull)
            order[t0 + n_front + bm] = pos;
        else
            order[t0 + n_front + tot_multi + (pos - bf - bm)] = pos;
    }
}

__global__ __launch_bounds__(256) void big_gather(BigDims dm, const int32_t *__restrict__ u, const int32_t *__restrict__ order,
                                                  const int4 *__restrict__ i_tmp, const int4 *__restrict__ j_tmp,
                                                  int2 *__restrict__ slot_pu, int4 *__restrict__ i_desc, int4 *__restrict__ j_desc)
{
    const int b = (int)blockIdx.y, nb = dm.nb_of(b);
    for (int sl = (int)(blockIdx.x * blockDim.x + threadIdx.x); sl < nb; sl += (int)(gridDim.x * blockDim.x)) {
        const int64_t t0 = (int64_t)b * dm.B, t = t0 + sl;
        const int pos = order[t];
        slot_pu[t] = make_int2(pos, u[t0 + pos]);
        i_desc[t] = i_tmp[t0 + pos];
        j_desc[t] = j_tmp[t0 + pos];
    }
}

// ---- the row-sharded step's two index builds (shard_kernels.h: shard_plan_batches, shard_owner_index) for batches of any size:
// the same outputs, bit for bit, from the same sort.  Arrays are flat [nseg][stride]; segment b holds segn[b] valid elements.
__global__ __launch_bounds__(256) void plan_fill(const int32_t *__restrict__ i, const int32_t *__restrict__ j, BigDims dm, int32_t G, int64_t S,
                                                 uint32_t *__restrict__ keys, uint32_t *__restrict__ vals, int64_t *__restrict__ seg0,
                                                 int32_t *__restrict__ segn)
{
    const int b = (int)blockIdx.y, nb = dm.nb_of(b);
    const int64_t t0 = (int64_t)b * dm.B, e0 = 2 * t0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        seg0[b] = e0;
        segn[b] = 2 * nb;
    }
    for (int k = (int)(blockIdx.x * blockDim.x + threadIdx.x); k < 2 * nb; k += (int)(gridDim.x * blockDim.x)) {
        const int64_t g = k < nb ? i[t0 + k] : j[t0 + k - nb];
        keys[e0 + k] = shard_owner_key(g, G, S);
        vals[e0 + k] = (uint32_t)k;
    }
}

// heads[e] = 1 where a run of equal keys starts (valid elements only); the element behind the last segment closes the scan
__global__ __launch_bounds__(256) void flat_heads(const uint32_t *__restrict__ keys, const int32_t *__restrict__ segn, int stride, int nseg,
                                                  int32_t *__restrict__ heads)
{
    const int b = (int)blockIdx.y, n = segn[b];
    const int64_t e0 = (int64_t)b * stride;
    if (b == 0 && blockIdx.x == 0 && threadIdx.x == 0)
        heads[(int64_t)nseg * stride] = 0;
    for (int k = (int)(blockIdx.x * blockDim.x + threadIdx.x); k < stride; k += (int)(gridDim.x * blockDim.x))
        heads[e0 + k] = (k < n && (k == 0 || keys[e0 + k] != keys[e0 + k - 1])) ? 1 : 0;
}

__global__ __launch_bounds__(256) void plan_emit(BigDims dm, int32_t G, int64_t S, const uint32_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                 const int32_t *__restrict__ heads, const int32_t *__restrict__ hs, int32_t *__restrict__ req_keys,
                                                 int32_t *__restrict__ n_req, int32_t *__restrict__ req_off, int32_t *__restrict__ ci,
                                                 int32_t *__restrict__ cj)
{
    const int b = (int)blockIdx.y, nb = dm.nb_of(b), n = 2 * nb;
    const int64_t t0 = (int64_t)b * dm.B, e0 = 2 * t0;
    const int base = hs[e0];
    if (blockIdx.x == 0 && (int)threadIdx.x <= G) { // owner o's requests start at the rank of the first unique key of an owner >= o
        const int o = (int)threadIdx.x;
        const int lb = o == G ? n : big_bound<false>(keys + e0, 0, n, (uint32_t)((int64_t)o * S));
        const int at = hs[e0 + lb] - base;
        req_off[(int64_t)b * (G + 1) + o] = at;
        if (o == G)
            n_req[b] = at;
    }
    for (int k = (int)(blockIdx.x * blockDim.x + threadIdx.x); k < n; k += (int)(gridDim.x * blockDim.x)) {
        const int64_t e = e0 + k;
        const int h = heads[e], r = hs[e] + h - 1 - base; // the rank of this slot's item among the batch's unique items
        if (h)
            req_keys[e0 + r] = (int32_t)keys[e];
        const int slot = (int)vals[e];
        if (slot < nb)
            ci[t0 + slot] = r;
        else
            cj[t0 + slot - nb] = r;
    }
}

__global__ __launch_bounds__(256) void own_fill(const int32_t *__restrict__ ids_recv, const int64_t *__restrict__ srv_base,
                                                const int32_t *__restrict__ srv_cnt, int32_t G, int stride, uint32_t *__restrict__ keys,
                                                uint32_t *__restrict__ vals, int32_t *__restrict__ srv_rows, int64_t *__restrict__ seg0,
                                                int32_t *__restrict__ segn)
{
    __shared__ int pre[66]; // slots of the peers before p
    const int b = (int)blockIdx.y;
    if (threadIdx.x == 0) {
        int at = 0;
        for (int p = 0; p < G; ++p) {
            pre[p] = at;
            at += srv_cnt[(int64_t)b * G + p];
        }
        pre[G] = at;
    }
    __syncthreads();
    const int n = pre[G];
    const int64_t e0 = (int64_t)b * stride;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        seg0[b] = e0;
        segn[b] = n;
    }
    for (int s = (int)(blockIdx.x * blockDim.x + threadIdx.x); s < n; s += (int)(gridDim.x * blockDim.x)) {
        int p = 0;
        while (p + 1 < G && s >= pre[p + 1])
            ++p;
        const int32_t row = ids_recv[srv_base[(int64_t)b * G + p] + (s - pre[p])];
        keys[e0 + s] = (uint32_t)row;
        vals[e0 + s] = (uint32_t)s;
        srv_rows[e0 + s] = row;
    }
}

__global__ __launch_bounds__(256) void own_emit(const int32_t *__restrict__ segn, int stride, const uint32_t *__restrict__ keys,
                                                const uint32_t *__restrict__ vals, const int32_t *__restrict__ heads, const int32_t *__restrict__ hs,
                                                int32_t *__restrict__ order, int32_t *__restrict__ urows, int32_t *__restrict__ useg,
                                                int32_t *__restrict__ n_urows)
{
    const int b = (int)blockIdx.y, n = segn[b];
    const int64_t e0 = (int64_t)b * stride;
    const int base = hs[e0];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int total = hs[e0 + n] - base;
        useg[(int64_t)b * (stride + 1) + total] = n;
        n_urows[b] = total;
    }
    for (int k = (int)(blockIdx.x * blockDim.x + threadIdx.x); k < n; k += (int)(gridDim.x * blockDim.x)) {
        const int64_t e = e0 + k;
        if (heads[e]) {
            const int r = hs[e] - base;
            urows[e0 + r] = (int32_t)keys[e];
            useg[(int64_t)b * (stride + 1) + r] = k;
        }
        order[e] = (int32_t)vals[e];
    }
}

inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }

} // namespace

int crec_lazy_index_big(crec_ctx *ctx, const int32_t *u, const int32_t *i, const int32_t *j, int64_t n_chunk, int32_t B, int64_t nbc,
                        int bits_u, int bits_q, int spw, const LazyIndexOut &out, hipStream_t st, int64_t reserve_nbc)
{
    CREC_REQUIRE(ctx, B >= 1 && B <= LAZY_MAX_B, "batch_size %d: the lazy step's descriptors hold batches of at most %d triples", B, LAZY_MAX_B);
    const int64_t E = 3 * (int64_t)B * nbc, NB = (int64_t)B * nbc;
    CREC_REQUIRE(ctx, nbc >= 1 && E + 1 < ((int64_t)1 << 31), "chunk of %lld batches of %d: too large for one index build", (long long)nbc, B);
    BigDims dm{n_chunk, B, (int32_t)nbc};
    const int end_bit = bits_u > bits_q ? bits_u : bits_q;

    // scratch: keys / vals before and after the sort (the unsorted copies are dead after it and hold the runs' starts and
    // lengths), positive-slot counts, front flags and their scan, the positions' flags and their scan, the sort's digit counters
    struct Lay {
        size_t k0, v0, k1, v1, ci, fp, fg, wi, wj, pf, pg, hist, seg0, segn, tmp, bytes, scan32, scan64;
    };
    const int tiles_max = sort_tiles(2 * (int64_t)B);
    const auto layout = [&](int64_t nb_, Lay *L) -> hipError_t {
        const int64_t E_ = 3 * (int64_t)B * nb_, NB_ = (int64_t)B * nb_;
        hipError_t e = rocprim::exclusive_scan(nullptr, L->scan32, (const int32_t *)nullptr, (int32_t *)nullptr, (int32_t)0, (size_t)(E_ + 1),
                                               rocprim::plus<int32_t>(), st);
        if (e != hipSuccess)
            return e;
        e = rocprim::exclusive_scan(nullptr, L->scan64, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, 0ull, (size_t)(NB_ + 1),
                                    rocprim::plus<unsigned long long>(), st);
        if (e != hipSuccess)
            return e;
        size_t off = 0;
        const auto take = [&off](size_t bytes) {
            const size_t at = off;
            off += up256(bytes);
            return at;
        };
        L->k0 = take(4 * (size_t)(E_ + 1)), L->v0 = take(4 * (size_t)(E_ + 1)), L->k1 = take(4 * (size_t)(E_ + 1)), L->v1 = take(4 * (size_t)(E_ + 1));
        L->ci = take(4 * (size_t)(E_ + 1)), L->fp = take(4 * (size_t)(E_ + 1)), L->fg = take(4 * (size_t)(E_ + 1));
        L->wi = take(4 * (size_t)NB_), L->wj = take(4 * (size_t)NB_), L->pf = take(8 * (size_t)(NB_ + 1)), L->pg = take(8 * (size_t)(NB_ + 1));
        L->hist = take(4 * (size_t)(2 * nb_) * 256 * tiles_max);
        L->seg0 = take(8 * (size_t)(2 * nb_)), L->segn = take(4 * (size_t)(2 * nb_));
        L->tmp = take((L->scan32 > L->scan64 ? L->scan32 : L->scan64) + 256);
        L->bytes = off;
        return hipSuccess;
    };
    Lay L{}, Lmax{};
    CREC_HIP(ctx, layout(nbc, &L));
    CREC_HIP(ctx, layout(reserve_nbc > nbc ? reserve_nbc : nbc, &Lmax)); // (a caller that pipelines chunks of growing size reserves for the largest up front)
    if (int rc = crec_reserve(ctx, &ctx->big_ws, Lmax.bytes))
        return rc;
    size_t scan32_tmp = L.scan32, scan64_tmp = L.scan64;
    const size_t o_k0 = L.k0, o_v0 = L.v0, o_k1 = L.k1, o_v1 = L.v1, o_ci = L.ci, o_fp = L.fp, o_fg = L.fg, o_wi = L.wi, o_wj = L.wj, o_pf = L.pf,
                 o_pg = L.pg, o_hist = L.hist, o_tmp = L.tmp;
    char *base = (char *)ctx->big_ws.ptr;
    uint32_t *ka = (uint32_t *)(base + o_k0), *va = (uint32_t *)(base + o_v0), *kb = (uint32_t *)(base + o_k1), *vb = (uint32_t *)(base + o_v1);
    int32_t *hist = (int32_t *)(base + o_hist);
    int32_t *rci = (int32_t *)(base + o_ci), *fp = (int32_t *)(base + o_fp), *fg = (int32_t *)(base + o_fg);
    int32_t *wi = (int32_t *)(base + o_wi), *wj = (int32_t *)(base + o_wj);
    unsigned long long *pf = (unsigned long long *)(base + o_pf), *pg = (unsigned long long *)(base + o_pg);
    void *tmp = base + o_tmp;
    SegSort sd{(const int64_t *)(base + L.seg0), (const int32_t *)(base + L.segn), tiles_max};

    // grid.y = the batch, grid.x = enough workgroups for `per_batch` elements of it (capped: the kernels stride)
    const auto blocks = [nbc](int64_t per_batch) {
        int64_t g = (per_batch + 255) / 256;
        g = g < 1 ? 1 : (g > 1024 ? 1024 : g);
        return dim3((unsigned)g, (unsigned)nbc);
    };
    hipLaunchKernelGGL(big_fill, blocks(3 * (int64_t)B), dim3(256), 0, st, u, i, j, dm, ka, va, (int64_t *)(base + L.seg0), (int32_t *)(base + L.segn));
    uint32_t *k1 = nullptr, *v1 = nullptr;
    seg_sort(sd, (int)(2 * nbc), end_bit, ka, va, kb, vb, hist, st, &k1, &v1);
    // (the pair of arrays the sort did not end in is free from here on: the runs' starts and lengths)
    int32_t *rstart = (int32_t *)(k1 == ka ? kb : ka), *rcnt = (int32_t *)(v1 == va ? vb : va);
    hipLaunchKernelGGL(big_runs, blocks(3 * (int64_t)B), dim3(256), 0, st, dm, (const uint32_t *)k1, (const uint32_t *)v1, rstart, rcnt, rci, fp, out.u_desc);
    CREC_HIP(ctx, rocprim::exclusive_scan(tmp, scan32_tmp, (const int32_t *)fp, fg, (int32_t)0, (size_t)(E + 1), rocprim::plus<int32_t>(), st));
    hipLaunchKernelGGL(big_item_desc, blocks(2 * (int64_t)B), dim3(256), 0, st, dm, spw, (const uint32_t *)k1, (const uint32_t *)v1, (const int32_t *)rstart,
                       (const int32_t *)rcnt, (const int32_t *)rci, (const int32_t *)fg, out.i_tmp, out.j_tmp, out.order, wi, wj);
    hipLaunchKernelGGL(big_part_flags, blocks(B), dim3(256), 0, st, dm, (const int32_t *)wi, (const int32_t *)wj, pf);
    CREC_HIP(ctx, rocprim::exclusive_scan(tmp, scan64_tmp, (const unsigned long long *)pf, pg, 0ull, (size_t)(NB + 1),
                                          rocprim::plus<unsigned long long>(), st));
    hipLaunchKernelGGL(big_emit_order, blocks(B), dim3(256), 0, st, dm, (const unsigned long long *)pg, out.order, out.nslots);
    hipLaunchKernelGGL(big_gather, blocks(B), dim3(256), 0, st, dm, u, (const int32_t *)out.order, (const int4 *)out.i_tmp, (const int4 *)out.j_tmp,
                       out.slot_pu, out.i_desc, out.j_desc);
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}

namespace {
// scratch of the two builds above: two pairs of sort arrays, head flags and their scan, the sort's counters, the segment table
struct FlatWs {
    uint32_t *ka, *va, *kb, *vb;
    int32_t *heads, *hs, *hist, *segn;
    int64_t *seg0;
    void *tmp;
    size_t scan_tmp;
    int tiles_max;
};
int flat_ws(crec_ctx *ctx, int64_t nseg, int64_t stride, hipStream_t st, FlatWs *w)
{
    const int64_t N = nseg * stride;
    CREC_REQUIRE(ctx, N + 1 < ((int64_t)1 << 31) && nseg <= 65535, "chunk of %lld batches of %lld slots: too large for one index build", (long long)nseg,
                 (long long)stride);
    w->tiles_max = sort_tiles(stride);
    w->scan_tmp = 0;
    CREC_HIP(ctx, rocprim::exclusive_scan(nullptr, w->scan_tmp, (const int32_t *)nullptr, (int32_t *)nullptr, (int32_t)0, (size_t)(N + 1),
                                          rocprim::plus<int32_t>(), st));
    size_t off = 0;
    const auto take = [&off](size_t bytes) {
        const size_t at = off;
        off += up256(bytes);
        return at;
    };
    const size_t o_ka = take(4 * (size_t)(N + 1)), o_va = take(4 * (size_t)(N + 1)), o_kb = take(4 * (size_t)(N + 1)), o_vb = take(4 * (size_t)(N + 1));
    const size_t o_h = take(4 * (size_t)(N + 1)), o_hs = take(4 * (size_t)(N + 1)), o_hist = take(4 * (size_t)nseg * 256 * w->tiles_max);
    const size_t o_s0 = take(8 * (size_t)nseg), o_sn = take(4 * (size_t)nseg), o_tmp = take(w->scan_tmp + 256);
    if (int rc = crec_reserve(ctx, &ctx->big_ws, off))
        return rc;
    char *base = (char *)ctx->big_ws.ptr;
    w->ka = (uint32_t *)(base + o_ka), w->va = (uint32_t *)(base + o_va), w->kb = (uint32_t *)(base + o_kb), w->vb = (uint32_t *)(base + o_vb);
    w->heads = (int32_t *)(base + o_h), w->hs = (int32_t *)(base + o_hs), w->hist = (int32_t *)(base + o_hist);
    w->seg0 = (int64_t *)(base + o_s0), w->segn = (int32_t *)(base + o_sn), w->tmp = base + o_tmp;
    return CREC_OK;
}
inline dim3 flat_blocks(int64_t per_seg, int64_t nseg)
{
    int64_t g = (per_seg + 255) / 256;
    g = g < 1 ? 1 : (g > 1024 ? 1024 : g);
    return dim3((unsigned)g, (unsigned)nseg);
}
} // namespace

// the scratch of crec_shard_owner_index_big for this shape, reserved ahead (a rank must know that it can serve a chunk BEFORE the
// ranks agree to go on: what an owner is asked for differs from rank to rank)
int crec_shard_owner_index_big_reserve(crec_ctx *ctx, int64_t nbc, int64_t stride, hipStream_t st)
{
    FlatWs w{};
    return flat_ws(ctx, nbc, stride, st, &w);
}

int crec_shard_plan_big(crec_ctx *ctx, const int32_t *i, const int32_t *j, int64_t n_chunk, int32_t B, int64_t nbc, int32_t G, int64_t S, int key_bits,
                        int32_t *req_keys, int32_t *n_req, int32_t *req_off, int32_t *ci, int32_t *cj, hipStream_t st)
{
    CREC_REQUIRE(ctx, B >= 1 && B <= LAZY_MAX_B && nbc >= 1 && G >= 1 && G <= 64, "shape");
    FlatWs w{};
    if (int rc = flat_ws(ctx, nbc, 2 * (int64_t)B, st, &w))
        return rc;
    const BigDims dm{n_chunk, B, (int32_t)nbc};
    const int stride = 2 * B;
    hipLaunchKernelGGL(plan_fill, flat_blocks(stride, nbc), dim3(256), 0, st, i, j, dm, G, S, w.ka, w.va, w.seg0, w.segn);
    uint32_t *ks = nullptr, *vs = nullptr;
    seg_sort(SegSort{w.seg0, w.segn, w.tiles_max}, (int)nbc, key_bits, w.ka, w.va, w.kb, w.vb, w.hist, st, &ks, &vs);
    hipLaunchKernelGGL(flat_heads, flat_blocks(stride, nbc), dim3(256), 0, st, (const uint32_t *)ks, (const int32_t *)w.segn, stride, (int)nbc, w.heads);
    CREC_HIP(ctx, rocprim::exclusive_scan(w.tmp, w.scan_tmp, (const int32_t *)w.heads, w.hs, (int32_t)0, (size_t)(nbc * stride + 1), rocprim::plus<int32_t>(), st));
    hipLaunchKernelGGL(plan_emit, flat_blocks(stride, nbc), dim3(256), 0, st, dm, G, S, (const uint32_t *)ks, (const uint32_t *)vs, (const int32_t *)w.heads,
                       (const int32_t *)w.hs, req_keys, n_req, req_off, ci, cj);
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}

int crec_shard_owner_index_big(crec_ctx *ctx, const int32_t *ids_recv, const int64_t *srv_base, const int32_t *srv_cnt, int32_t G, int64_t nbc,
                               int64_t stride, int key_bits, int32_t *srv_rows, int32_t *order, int32_t *urows, int32_t *useg, int32_t *n_urows,
                               hipStream_t st)
{
    CREC_REQUIRE(ctx, nbc >= 1 && G >= 1 && G <= 64 && stride >= 1 && stride <= 2 * (int64_t)LAZY_MAX_B * 64, "shape");
    FlatWs w{};
    if (int rc = flat_ws(ctx, nbc, stride, st, &w))
        return rc;
    hipLaunchKernelGGL(own_fill, flat_blocks(stride, nbc), dim3(256), 0, st, ids_recv, srv_base, srv_cnt, G, (int)stride, w.ka, w.va, srv_rows, w.seg0, w.segn);
    uint32_t *ks = nullptr, *vs = nullptr;
    seg_sort(SegSort{w.seg0, w.segn, w.tiles_max}, (int)nbc, key_bits, w.ka, w.va, w.kb, w.vb, w.hist, st, &ks, &vs);
    hipLaunchKernelGGL(flat_heads, flat_blocks(stride, nbc), dim3(256), 0, st, (const uint32_t *)ks, (const int32_t *)w.segn, (int)stride, (int)nbc, w.heads);
    CREC_HIP(ctx, rocprim::exclusive_scan(w.tmp, w.scan_tmp, (const int32_t *)w.heads, w.hs, (int32_t)0, (size_t)(nbc * stride + 1), rocprim::plus<int32_t>(), st));
    hipLaunchKernelGGL(own_emit, flat_blocks(stride, nbc), dim3(256), 0, st, (const int32_t *)w.segn, (int)stride, (const uint32_t *)ks, (const uint32_t *)vs,
                       (const int32_t *)w.heads, (const int32_t *)w.hs, order, urows, useg, n_urows);
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}
