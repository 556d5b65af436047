// Full-catalogue scoring of MLP / NeuMF (and GMF) and top-K over a materialised score matrix.
//
// Replaces sess.run(self.pre_scores) of the rs branch of MLP._predict / NeuMF._predict
// (model/ranking/MLP.py:45-52,71-77, NeuMF.py:62-78,96-105: every test user against every item through
// the tower) and the np.argsort + seen-filter loop (model/RankingRecommender.py:221-240).
//
// The first tower layer is linear in concat(pm_u, qm_i): (pm_u W0[:h]) + (qm_i W0[h:] + b0), so it is
// computed once per user and once per item (ncf_first_layer) and a pair only adds the two and applies ReLU.
// ncf_pair_logits: one thread per (user, item) pair, activations in LDS ([k][thread]: conflict-free),
// weights read with wave-uniform addresses.  The logit, not its sigmoid, is written (same ranking).
// topk_rows: one wave per user over a row of the score matrix staged in LDS: seen items are struck out,
// then K rounds of wave-wide arg-max (ties: lower item id).
#include "crec_internal.h"
#include "ncf_common.h"

namespace {

constexpr int PT = 256; // pairs (threads) per workgroup: 4 users x 64 items
constexpr int TI = 64;

// out[r][o] = sum_k X[row_id(r)][k] * W[(k0 + k)][o] (+ b[o]);  rows = users (gathered) or all items
__global__ void ncf_first_layer(const float *__restrict__ X, const int32_t *__restrict__ ids, int64_t rows, int hm,
                                const float *__restrict__ W, int k0, int kout, const float *__restrict__ b,
                                float *__restrict__ out)
{
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= rows * kout)
        return;
    const int64_t r = gid / kout;
    const int o = (int)(gid % kout);
    const float *x = X + (ids ? (int64_t)ids[r] : r) * hm;
    float s = b ? b[o] : 0.f;
    for (int k = 0; k < hm; ++k)
        s = s + x[k] * W[(int64_t)(k0 + k) * kout + o];
    out[gid] = s;
}

__global__ __launch_bounds__(PT) void ncf_pair_logits(NcfDev m, const int32_t *__restrict__ users, int Bt, int64_t I,
                                                      const float *__restrict__ Au, const float *__restrict__ Bi,
                                                      float *__restrict__ logits)
{
    extern __shared__ float lds[]; // two activation buffers [width][PT]
    const int tid = threadIdx.x;
    const int ul = tid / TI, il = tid % TI;
    const int64_t ur = (int64_t)blockIdx.y * (PT / TI) + ul, it = (int64_t)blockIdx.x * TI + il;
    const bool ok = ur < Bt && it < I;
    const bool has_g = m.kind != 1, has_m = m.kind != 0;
    const int dg = has_g ? m.dg : 0;
    const float *h = m.small + m.h_off;
    float z = 0.f;
    if (has_g && ok) {
        const float *pg = m.Pg + (int64_t)users[ur] * dg, *qg = m.Qg + it * dg;
        for (int c = 0; c < dg; ++c)
            z = z + (pg[c] * qg[c]) * h[c];
    }
    if (has_m) {
        int w = m.L0 / 2; // width after the first layer
        float *cur = lds, *nxt = lds + (size_t)w * PT;
        for (int k = 0; k < w; ++k) {
            const float v = ok ? Au[ur * w + k] + Bi[it * w + k] : 0.f;
            cur[k * PT + tid] = v > 0.f ? v : 0.f;
        }
        for (int layer = 1; layer < m.n_layers; ++layer) {
            const int kout = w / 2;
            const float *W = m.small + m.w_off[layer], *b = m.small + m.b_off[layer];
            for (int o = 0; o < kout; ++o) {
                float s = b[o];
                for (int k = 0; k < w; ++k)
                    s = s + cur[k * PT + tid] * W[k * kout + o];
                nxt[o * PT + tid] = s > 0.f ? s : 0.f;
            }
            float *t = cur;
            cur = nxt;
            nxt = t;
            w = kout;
        }
        const float *h_m = h + dg;
        for (int k = 0; k < w; ++k)
            z = z + cur[k * PT + tid] * h_m[k];
    }
    if (ok)
        logits[ur * I + it] = z;
}

typedef unsigned long long u64;
__device__ inline uint32_t ordered_bits(float f)
{
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}

// one wave per user row; the row is staged in LDS ([I] floats), seen items struck out, K arg-max rounds
__global__ __launch_bounds__(64) void topk_rows(const float *__restrict__ scores, int64_t I, const int32_t *__restrict__ users,
                                                const int64_t *__restrict__ seen_ptr,
                                                const int32_t *__restrict__ seen_items, int K, float sign,
                                                int32_t *__restrict__ ids, float *__restrict__ vals)
{
    extern __shared__ float row[];
    const int lane = threadIdx.x;
    const int64_t r = blockIdx.x;
    const float worst = -INFINITY;
    const bool keep_listed = K >> 16; // the host packs the filter mode above K
    K &= 0xffff;
    const int32_t uu = users[r];
    if (!keep_listed) { // listed = seen in training: struck out
        for (int64_t c = lane; c < I; c += 64)
            row[c] = sign * scores[r * I + c];
        __syncthreads();
        for (int64_t e = seen_ptr[uu] + lane; e < seen_ptr[uu + 1]; e += 64)
            row[seen_items[e]] = worst;
    } else { // listed = the user's candidates: only they compete
        for (int64_t c = lane; c < I; c += 64)
            row[c] = worst;
        __syncthreads();
        for (int64_t e = seen_ptr[uu] + lane; e < seen_ptr[uu + 1]; e += 64)
            row[seen_items[e]] = sign * scores[r * I + seen_items[e]];
    }
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        u64 best = 0; // (ordered score, ~item): larger is better, ties -> lower item id
        for (int64_t c = lane; c < I; c += 64) {
            const float s = row[c];
            if (s != worst) {
                const u64 key = ((u64)ordered_bits(s) << 32) | (0xffffffffu - (uint32_t)c);
                best = key > best ? key : best;
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const u64 o = __shfl_xor(best, off, 64);
            best = o > best ? o : best;
        }
        if (best == 0) {
            if (lane == 0) {
                ids[r * K + k] = -1;
                vals[r * K + k] = sign * worst;
            }
        } else {
            const int item = (int)(0xffffffffu - (uint32_t)best);
            if (lane == 0) {
                ids[r * K + k] = item;
                vals[r * K + k] = sign * row[item];
            }
            __syncthreads();
            if (lane == 0)
                row[item] = worst;
        }
        __syncthreads();
    }
}

} // namespace

extern "C" {

int crec_ncf_scores(crec_ctx *ctx, const crec_ncf *mdl, const int32_t *users, int32_t Bt, float *logits, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, mdl && users && logits && Bt >= 0, "null argument");
    NcfDev d{};
    if (int rc = ncf_describe(ctx, mdl, &d))
        return rc;
    if (Bt == 0)
        return CREC_OK;
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const bool has_m = d.kind != CREC_NCF_GMF;
    const int64_t I = has_m ? mdl->Q_mlp.rows : mdl->Q_gmf.rows;
    const int hm = d.L0 / 2, w1 = d.L0 / 2; // first layer maps L0 -> L0/2
    float *Au = nullptr, *Bi = nullptr;
    size_t lds = 0;
    if (has_m) {
        auto up = [](size_t x) { return (x + 255) / 256 * 256; };
        const size_t ab = up(sizeof(float) * (size_t)Bt * w1), bb = up(sizeof(float) * (size_t)I * w1);
        if (int rc = crec_reserve(ctx, &ctx->eval_ws, ab + bb))
            return rc;
        Au = (float *)ctx->eval_ws.ptr;
        Bi = (float *)((char *)ctx->eval_ws.ptr + ab);
        const float *W0 = d.small + d.w_off[0], *b0 = d.small + d.b_off[0];
        hipLaunchKernelGGL(ncf_first_layer, dim3((unsigned)crec_ceil_div((int64_t)Bt * w1, 256)), dim3(256), 0, st, d.Pm, users,
                           (int64_t)Bt, hm, W0, 0, w1, (const float *)nullptr, Au);
        hipLaunchKernelGGL(ncf_first_layer, dim3((unsigned)crec_ceil_div(I * w1, 256)), dim3(256), 0, st, d.Qm,
                           (const int32_t *)nullptr, I, hm, W0, hm, w1, b0, Bi);
        lds = sizeof(float) * (size_t)(w1 + w1 / 2) * PT;
        CREC_REQUIRE(ctx, lds <= 160 * 1024, "tower too wide for the pair kernel (layers[0] <= 256)");
        CREC_HIP(ctx, hipFuncSetAttribute((const void *)ncf_pair_logits, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const dim3 grid((unsigned)crec_ceil_div(I, TI), (unsigned)crec_ceil_div(Bt, PT / TI));
    crec_prof_begin(ctx, CREC_K_NCF_SCORES, st);
    CREC_KLAUNCH(ctx, ncf_pair_logits, grid, dim3(PT), lds, st, d, users, (int)Bt, I, (const float *)Au, (const float *)Bi,
                       logits);
    crec_prof_end(ctx, st);
    CREC_HIP(ctx, hipGetLastError());
    return crec_prof_flush(ctx, st);
}

static int topk_scores_impl(crec_ctx *ctx, const float *scores, int64_t I, const int32_t *users, int32_t Bt,
                            const int64_t *seen_ptr, const int32_t *seen_items, int32_t K, int32_t ascending,
                            int32_t *topk_ids, float *topk_scores, void *stream, int keep_listed);

int crec_topk_scores(crec_ctx *ctx, const float *scores, int64_t I, const int32_t *users, int32_t Bt,
                     const int64_t *seen_ptr, const int32_t *seen_items, int32_t K, int32_t ascending, int32_t *topk_ids,
                     float *topk_scores, void *stream)
{
    return topk_scores_impl(ctx, scores, I, users, Bt, seen_ptr, seen_items, K, ascending, topk_ids, topk_scores, stream, 0);
}

int crec_topk_scores_candidates(crec_ctx *ctx, const float *scores, int64_t I, const int32_t *users, int32_t Bt,
                                const int64_t *cand_ptr, const int32_t *cand_items, int32_t K, int32_t ascending,
                                int32_t *topk_ids, float *topk_scores, void *stream)
{
    return topk_scores_impl(ctx, scores, I, users, Bt, cand_ptr, cand_items, K, ascending, topk_ids, topk_scores, stream, 1);
}

static int topk_scores_impl(crec_ctx *ctx, const float *scores, int64_t I, const int32_t *users, int32_t Bt,
                            const int64_t *seen_ptr, const int32_t *seen_items, int32_t K, int32_t ascending,
                            int32_t *topk_ids, float *topk_scores, void *stream, int keep_listed)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, scores && users && seen_ptr && topk_ids && topk_scores, "null argument");
    CREC_REQUIRE(ctx, I >= 1 && I <= 40000 && K >= 1 && Bt >= 0, "score rows of at most 40000 items are staged in LDS");
    if (Bt == 0)
        return CREC_OK;
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    const size_t lds = sizeof(float) * (size_t)I;
    CREC_HIP(ctx, hipFuncSetAttribute((const void *)topk_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(topk_rows, dim3((unsigned)Bt), dim3(64), lds, (hipStream_t)stream, scores, I, users, seen_ptr, seen_items,
                       (int)K | (keep_listed << 16), ascending ? -1.0f : 1.0f, topk_ids, topk_scores);
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}

} // extern "C"
