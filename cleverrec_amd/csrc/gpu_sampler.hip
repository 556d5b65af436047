// Device-side negative sampling: the statistically equivalent, NOT stream-compatible, form of
// utils/sampler.py:10-99 for runs that do not need NumPy's MT19937 stream reproduced.
//
// The reference draws, for every training positive (u, i), neg_ratio items uniformly from the catalogue, redrawing
// while the item is one of u's training items or already in this positive's group (sampler.py:18-31, 50-66, 82-93).
// That rule is kept exactly; what changes is the generator: a counter-based hash of (seed, positive, draw index)
// instead of one serial MT19937 stream, so every positive is independent and the epoch is drawn by one launch
// (one thread per positive).  The epoch shuffle (np.random.permutation + fancy indexing) is a device randperm +
// gathers on the host side of this ABI (cleverrec_amd/runtime.py).  The bit-exact host samplers (crec_sampler.cpp)
// stay the default and the parity oracle; see DESIGN.md 4.6.
#include "crec_internal.h"

namespace {

// two rounds of the splitmix64 finaliser over a 128-bit counter: (seed, epoch) x (positive, draw)
__device__ inline uint32_t hash_draw(uint64_t key, uint64_t e, uint32_t n)
{
    uint64_t z = key + 0x9e3779b97f4a7c15ull * (e + 1) + 0xbf58476d1ce4e5b9ull * (uint64_t)(n + 1);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z = z ^ (z >> 31);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return (uint32_t)((z ^ (z >> 31)) >> 16);
}

// one thread per positive e: user = edge_user[e]; its neg_ratio negatives -> neg[e][0..neg_ratio)
__global__ __launch_bounds__(256) void draw_negatives_device(uint64_t key, int64_t n_pos, const int32_t *__restrict__ edge_user,
                                                             const int64_t *__restrict__ seen_ptr,
                                                             const int32_t *__restrict__ seen_items, int32_t item_nums,
                                                             int neg_ratio, int32_t *__restrict__ neg)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_pos)
        return;
    const int32_t u = edge_user[e];
    const int64_t s0 = seen_ptr[u], s1 = seen_ptr[u + 1];
    const uint32_t max = (uint32_t)(item_nums - 1);
    uint32_t mask = max; // smallest 2^k - 1 >= max: masked rejection keeps the draw exactly uniform
    mask |= mask >> 1, mask |= mask >> 2, mask |= mask >> 4, mask |= mask >> 8, mask |= mask >> 16;
    int32_t *grp = neg + e * neg_ratio;
    uint32_t n = 0;
    for (int t = 0; t < neg_ratio; ++t) {
        for (;;) {
            const uint32_t v = hash_draw(key, (uint64_t)e, n++) & mask;
            if (v > max)
                continue;
            const int32_t j = (int32_t)v;
            int64_t lo = s0, hi = s1; // j in the user's sorted training items?
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (seen_items[mid] < j)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            bool bad = lo < s1 && seen_items[lo] == j;
            for (int q = 0; q < t; ++q)
                bad |= grp[q] == j;
            if (!bad) {
                grp[t] = j;
                break;
            }
        }
    }
}

} // namespace

extern "C" int crec_sample_negatives_device(crec_ctx *ctx, uint64_t seed, uint64_t epoch, int64_t n_pos,
                                            const int32_t *edge_user, const int64_t *seen_ptr, const int32_t *seen_items,
                                            const int32_t *max_seen, int32_t item_nums, int32_t neg_ratio, int32_t *neg,
                                            void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, n_pos >= 0 && (n_pos == 0 || (edge_user && seen_ptr && seen_items && neg)), "null argument");
    CREC_REQUIRE(ctx, item_nums >= 1 && neg_ratio >= 1 && neg_ratio <= 1024, "item_nums >= 1, 1 <= neg_ratio <= 1024");
    // the rejection loop terminates only if every user can supply neg_ratio unseen items (the host sampler checks the
    // same and fails with CREC_EINVAL); the caller passes the largest number of distinct training items of a user
    CREC_REQUIRE(ctx, max_seen && *max_seen + neg_ratio <= item_nums, "a user has fewer than neg_ratio unseen items");
    if (n_pos == 0)
        return CREC_OK;
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t key = (seed * 0xd1342543de82ef95ull) ^ (epoch * 0x2545f4914f6cdd1dull + 0x632be59bd9b4e019ull);
    hipLaunchKernelGGL(draw_negatives_device, dim3((unsigned)crec_ceil_div(n_pos, 256)), dim3(256), 0, (hipStream_t)stream, key,
                       n_pos, edge_user, seen_ptr, seen_items, item_nums, (int)neg_ratio, neg);
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}
