// index_big.hip: the lazy step's index for batches of ANY size (the reference takes any `batch_size`: model/Recommender.py:17).
//
// bpr_index_batches sorts a batch's (row, slot) pairs in LDS -- at most 16 384 pairs, batches of at most 8 192 triples.  Here the
// same index is built in global memory for a whole chunk of batches at once, every phase a flat launch over all of the chunk's
// elements (no workgroup owns a batch, so nothing grows with the batch):
//
//   big_fill         (row, slot) pairs of both sides of every batch; per batch a user segment and an item segment (i-slots, then
//                    j-slots, so that a stable sort leaves a row's occurrences in batch order, positives first)
//   rocPRIM          segmented stable radix sort by row over the significant bits of a row id
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
__device__ __forceinline__ BigAt big_at(const BigDims &dm, int64_t e)
{
    BigAt a;
    const int64_t per = 3 * (int64_t)dm.B;
    a.b = (int)(e / per);
    const int r = (int)(e - (int64_t)a.b * per);
    a.nb = dm.nb_of(a.b);
    a.side = r >= dm.B ? 1 : 0;
    a.k = a.side ? r - dm.B : r;
    a.n = a.side ? 2 * a.nb : a.nb;
    a.seg = (int64_t)a.b * per + (a.side ? dm.B : 0);
    return a;
}

__global__ __launch_bounds__(256) void big_fill(const int32_t *__restrict__ u, const int32_t *__restrict__ i, const int32_t *__restrict__ j,
                                                BigDims dm, uint32_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                uint32_t *__restrict__ seg_begin, uint32_t *__restrict__ seg_end)
{
    const int64_t E = 3 * (int64_t)dm.B * dm.nbc;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const BigAt a = big_at(dm, e);
        if (e < 2 * (int64_t)dm.nbc) { // segment s = 2 * batch + side
            const int b = (int)(e >> 1), side = (int)(e & 1), nb = dm.nb_of(b);
            const int64_t s0 = 3 * (int64_t)dm.B * b + (side ? dm.B : 0);
            seg_begin[e] = (uint32_t)s0;
            seg_end[e] = (uint32_t)(s0 + (side ? 2 * nb : nb));
        }
        uint32_t key = 0xffffffffu;
        if (a.k < a.n) {
            const int64_t t0 = (int64_t)a.b * dm.B;
            key = (uint32_t)(a.side == 0 ? u[t0 + a.k] : (a.k < a.nb ? i[t0 + a.k] : j[t0 + a.k - a.nb]));
        }
        keys[e] = key;
        vals[e] = (uint32_t)a.k;
    }
}

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
    const int64_t E = 3 * (int64_t)dm.B * dm.nbc;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e <= E; e += (int64_t)gridDim.x * blockDim.x) {
        if (e == E) { // the scan's closing element
            fp[e] = 0;
            continue;
        }
        const BigAt a = big_at(dm, e);
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
    const int64_t n_items = 2 * (int64_t)dm.B * dm.nbc;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n_items; t += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(t / (2 * dm.B)), k = (int)(t - (int64_t)b * 2 * dm.B), nb = dm.nb_of(b);
        if (k >= 2 * nb)
            continue;
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
    const int64_t n = (int64_t)dm.B * dm.nbc;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t <= n; t += (int64_t)gridDim.x * blockDim.x) {
        unsigned long long f = 0;
        if (t < n) {
            const int b = (int)(t / dm.B), pos = (int)(t - (int64_t)b * dm.B);
            if (pos < dm.nb_of(b)) {
                const int a = wi[t], c = wj[t];
                if (a == -1)
                    f = 1ull << 32;
                else if ((a > c ? a : c) >= 2)
                    f = 1ull;
            }
        }
        pf[t] = f;
    }
}

__global__ __launch_bounds__(256) void big_emit_order(BigDims dm, const int32_t *__restrict__ wi, const unsigned long long *__restrict__ pg,
                                                      int32_t *__restrict__ order, int32_t *__restrict__ nslots)
{
    const int64_t n = (int64_t)dm.B * dm.nbc;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(t / dm.B), pos = (int)(t - (int64_t)b * dm.B), nb = dm.nb_of(b);
        if (pos >= nb)
            continue;
        const int64_t t0 = (int64_t)b * dm.B;
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
    (void)wi;
}

__global__ __launch_bounds__(256) void big_gather(BigDims dm, const int32_t *__restrict__ u, const int32_t *__restrict__ order,
                                                  const int4 *__restrict__ i_tmp, const int4 *__restrict__ j_tmp,
                                                  int2 *__restrict__ slot_pu, int4 *__restrict__ i_desc, int4 *__restrict__ j_desc)
{
    const int64_t n = (int64_t)dm.B * dm.nbc;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(t / dm.B), sl = (int)(t - (int64_t)b * dm.B);
        if (sl >= dm.nb_of(b))
            continue;
        const int64_t t0 = (int64_t)b * dm.B;
        const int pos = order[t];
        slot_pu[t] = make_int2(pos, u[t0 + pos]);
        i_desc[t] = i_tmp[t0 + pos];
        j_desc[t] = j_tmp[t0 + pos];
    }
}

inline size_t up256(size_t x) { return (x + 255) / 256 * 256; }

} // namespace

int crec_lazy_index_big(crec_ctx *ctx, const int32_t *u, const int32_t *i, const int32_t *j, int64_t n_chunk, int32_t B, int64_t nbc,
                        int bits_u, int bits_q, int spw, const LazyIndexOut &out, hipStream_t st)
{
    CREC_REQUIRE(ctx, B >= 1 && B <= LAZY_MAX_B, "batch_size %d: the lazy step's descriptors hold batches of at most %d triples", B, LAZY_MAX_B);
    const int64_t E = 3 * (int64_t)B * nbc, NB = (int64_t)B * nbc;
    CREC_REQUIRE(ctx, nbc >= 1 && E + 1 < ((int64_t)1 << 31), "chunk of %lld batches of %d: too large for one index build", (long long)nbc, B);
    BigDims dm{n_chunk, B, (int32_t)nbc};
    const int end_bit = bits_u > bits_q ? bits_u : bits_q;

    // scratch: keys / vals before and after the sort (the unsorted copies are dead after it and hold the runs' starts and
    // lengths), positive-slot counts, front flags and their scan, the positions' flags and their scan, segment offsets
    size_t sort_tmp = 0, scan32_tmp = 0, scan64_tmp = 0;
    CREC_HIP(ctx, rocprim::segmented_radix_sort_pairs(nullptr, sort_tmp, (const uint32_t *)nullptr, (uint32_t *)nullptr, (const uint32_t *)nullptr,
                                                      (uint32_t *)nullptr, (unsigned)E, (unsigned)(2 * nbc), (const uint32_t *)nullptr,
                                                      (const uint32_t *)nullptr, 0u, (unsigned)end_bit, st));
    CREC_HIP(ctx, rocprim::exclusive_scan(nullptr, scan32_tmp, (const int32_t *)nullptr, (int32_t *)nullptr, (int32_t)0, (size_t)(E + 1),
                                          rocprim::plus<int32_t>(), st));
    CREC_HIP(ctx, rocprim::exclusive_scan(nullptr, scan64_tmp, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, 0ull,
                                          (size_t)(NB + 1), rocprim::plus<unsigned long long>(), st));
    size_t tmp_bytes = sort_tmp > scan32_tmp ? sort_tmp : scan32_tmp;
    tmp_bytes = tmp_bytes > scan64_tmp ? tmp_bytes : scan64_tmp;
    size_t off = 0;
    const auto take = [&off](size_t bytes) {
        const size_t at = off;
        off += up256(bytes);
        return at;
    };
    const size_t o_k0 = take(4 * (size_t)(E + 1)), o_v0 = take(4 * (size_t)(E + 1)), o_k1 = take(4 * (size_t)(E + 1)), o_v1 = take(4 * (size_t)(E + 1));
    const size_t o_ci = take(4 * (size_t)(E + 1)), o_fp = take(4 * (size_t)(E + 1)), o_fg = take(4 * (size_t)(E + 1));
    const size_t o_wi = take(4 * (size_t)NB), o_wj = take(4 * (size_t)NB), o_pf = take(8 * (size_t)(NB + 1)), o_pg = take(8 * (size_t)(NB + 1));
    const size_t o_sb = take(4 * (size_t)(2 * nbc)), o_se = take(4 * (size_t)(2 * nbc)), o_tmp = take(tmp_bytes + 256);
    if (int rc = crec_reserve(ctx, &ctx->big_ws, off))
        return rc;
    char *base = (char *)ctx->big_ws.ptr;
    uint32_t *k0 = (uint32_t *)(base + o_k0), *v0 = (uint32_t *)(base + o_v0), *k1 = (uint32_t *)(base + o_k1), *v1 = (uint32_t *)(base + o_v1);
    int32_t *rstart = (int32_t *)k0, *rcnt = (int32_t *)v0; // (after the sort)
    int32_t *rci = (int32_t *)(base + o_ci), *fp = (int32_t *)(base + o_fp), *fg = (int32_t *)(base + o_fg);
    int32_t *wi = (int32_t *)(base + o_wi), *wj = (int32_t *)(base + o_wj);
    unsigned long long *pf = (unsigned long long *)(base + o_pf), *pg = (unsigned long long *)(base + o_pg);
    uint32_t *sb = (uint32_t *)(base + o_sb), *se = (uint32_t *)(base + o_se);
    void *tmp = base + o_tmp;

    const auto blocks = [](int64_t n) { const int64_t g = (n + 255) / 256; return dim3((unsigned)(g < 1 ? 1 : (g > 65536 ? 65536 : g))); };
    hipLaunchKernelGGL(big_fill, blocks(E), dim3(256), 0, st, u, i, j, dm, k0, v0, sb, se);
    CREC_HIP(ctx, rocprim::segmented_radix_sort_pairs(tmp, sort_tmp, (const uint32_t *)k0, k1, (const uint32_t *)v0, v1, (unsigned)E,
                                                      (unsigned)(2 * nbc), (const uint32_t *)sb, (const uint32_t *)se, 0u, (unsigned)end_bit, st));
    hipLaunchKernelGGL(big_runs, blocks(E + 1), dim3(256), 0, st, dm, (const uint32_t *)k1, (const uint32_t *)v1, rstart, rcnt, rci, fp, out.u_desc);
    CREC_HIP(ctx, rocprim::exclusive_scan(tmp, scan32_tmp, (const int32_t *)fp, fg, (int32_t)0, (size_t)(E + 1), rocprim::plus<int32_t>(), st));
    hipLaunchKernelGGL(big_item_desc, blocks(2 * NB), dim3(256), 0, st, dm, spw, (const uint32_t *)k1, (const uint32_t *)v1, (const int32_t *)rstart,
                       (const int32_t *)rcnt, (const int32_t *)rci, (const int32_t *)fg, out.i_tmp, out.j_tmp, out.order, wi, wj);
    hipLaunchKernelGGL(big_part_flags, blocks(NB + 1), dim3(256), 0, st, dm, (const int32_t *)wi, (const int32_t *)wj, pf);
    CREC_HIP(ctx, rocprim::exclusive_scan(tmp, scan64_tmp, (const unsigned long long *)pf, pg, 0ull, (size_t)(NB + 1),
                                          rocprim::plus<unsigned long long>(), st));
    hipLaunchKernelGGL(big_emit_order, blocks(NB), dim3(256), 0, st, dm, (const int32_t *)wi, (const unsigned long long *)pg, out.order, out.nslots);
    hipLaunchKernelGGL(big_gather, blocks(NB), dim3(256), 0, st, dm, u, (const int32_t *)out.order, (const int4 *)out.i_tmp, (const int4 *)out.j_tmp,
                       out.slot_pu, out.i_desc, out.j_desc);
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}
