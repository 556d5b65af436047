// Full-catalogue scoring of MLP / NeuMF (and GMF) and top-K over a materialised score matrix.
//
// Replaces sess.run(self.pre_scores) of the rs branch of MLP._predict / NeuMF._predict
// (model/ranking/MLP.py:45-52,71-77, NeuMF.py:62-78,96-105: every test user against every item through
// the tower) and the np.argsort + seen-filter loop (model/RankingRecommender.py:221-240).
//
// The first tower layer is linear in concat(pm_u, qm_i): (pm_u W0[:h]) + (qm_i W0[h:] + b0), so it is
// computed once per user and once per item (ncf_first_layer) and a pair only adds the two and applies ReLU.
// ncf_pair_logits: a workgroup per 4 users walking tiles of 64 items; the remaining layers run on the fp32 matrix cores
// between LDS activation buffers.  The logit, not its sigmoid, is written (same ranking).
// topk_rows: one workgroup per user over a row of the score matrix staged in LDS: seen items are struck out,
// then K rounds of arg-max over per-thread running bests (ties: lower item id).
#include "crec_internal.h"
#include "ncf_common.h"

namespace {

constexpr int PT = 512; // threads per workgroup (two waves per SIMD)
constexpr int TI = 64;  // items per tile
constexpr int TU = 4;   // users per tile (PT == 2 * TU * TI: two threads per pair in the GMF part)
constexpr int TP = TU * TI; // (user, item) pairs per tile

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

// A workgroup owns TU users and walks item tiles of TI rows: a tile is TP = TU x TI pairs, pair p = (p / TI, p % TI).
//   A_1[p] = relu(Au[user] + Bi[item]) (first layer, precomputed per user and per item) is never stored: it is formed
//   on the fly as the A operand of the next layer's product.  Every further layer is a [TP x w] x [w x w/2] product
//   on v_mfma_f32_16x16x4_f32 (tile_gemm, ncf_common.h) between LDS activation buffers.  Weights, biases, h and
//   the users' rows are staged in LDS once per workgroup; a tile's Bi / Q_gmf rows are prefetched into registers
//   while the previous tile is computed.  The GMF part is a dot product per pair against the tile's Q_gmf rows.
//   The logit, not its sigmoid, is written (same ranking).
constexpr int MAXF4 = 4; // float4s of a tile's [TI][width] rows per thread: width <= MAXF4 * 4 * PT / TI = 128

// <L0C, LC, DGC>: the tower shape fixed at compile time (the shipped confs' layers [128, 64, 32]; L0C < 0: from NcfDev), as
// for the training tile (csrc/ncf.hip): the tile is bound by its bookkeeping instructions, which fold away with the shape
template <int L0C, int LC, int DGC>
__global__ __launch_bounds__(PT) void ncf_pair_logits(NcfDev m, const int32_t *__restrict__ users, int Bt, int64_t I,
                                                      const float *__restrict__ Au, const float *__restrict__ Bi,
                                                      float *__restrict__ logits)
{
    extern __shared__ float lds[];
    constexpr int NWV = PT / 64;
    const int tid = threadIdx.x;
    const int64_t u0 = (int64_t)blockIdx.y * TU;
    constexpr bool FIXED = L0C >= 0;
    const bool has_g = FIXED ? DGC > 0 : m.kind != 1, has_m = FIXED ? LC > 0 : m.kind != 0;
    const int L0 = FIXED ? L0C : m.L0;
    const int dg = FIXED ? DGC : (has_g ? m.dg : 0), ldg = dg + 1;
    const int w1 = has_m ? L0 / 2 : 0, ldb = w1 + 1, L = FIXED ? LC : (has_m ? m.n_layers : 0);
    const auto width_of = [&](int k) { return L0 >> k; }; // input width of layer k
    const auto fw_of = [&](int k) { // flat offset of W_k in the parameter block [W_0 | b_0 | ... | h]; fw_of(L) = h's
        int o = 0;
        for (int j = 0; j < k; ++j)
            o += width_of(j) * width_of(j + 1) + width_of(j + 1);
        return o;
    };
    const auto woff_of = [&](int k) {                         // floats before W_k among layers 1..
        int o = 0;
        for (int j = 1; j < k; ++j)
            o += width_of(j) * (width_of(j + 1) + 1);
        return o;
    };
    const auto boff_of = [&](int k) {
        int o = 0;
        for (int j = 1; j < k; ++j)
            o += width_of(j + 1);
        return o;
    };
    const int last = has_m ? width_of(L) : 0;
    // carve (ncf_pair_lds_floats mirrors this)
    float *Bl = lds;                             // [TI][w1 + 1]   the tile's first-layer item halves
    float *Ql = Bl + TI * ldb;                   // [TI][dg + 1]   the tile's Q_gmf rows
    float *Aul = Ql + TI * ldg;                  // [TU][w1]       the users' first-layer halves
    float *ph = Aul + TU * w1;                   // [TU][dg]       p_gmf * h_gmf
    float *buf1 = ph + TU * dg;                  // [TP][w1 / 2 + 1]
    float *buf2 = buf1 + TP * (w1 / 2 + 1);      // [TP][w1 / 4 + 1]
    float *zg = buf2 + TP * (w1 / 4 + 1);        // [TP]
    float *b_base = zg + TP;
    float *hl = b_base + boff_of(L);             // h of the tower part [last]
    float *W_base = hl + last;

    // ---- once per workgroup: parameters and the users' rows ---------------------------------------------------------
    const float *h = m.small + (has_m ? fw_of(L) : 0);
    for (int c = tid; c < last; c += PT)
        hl[c] = h[dg + c];
    for (int k = 1; k < L; ++k) {
        for (int c = tid; c < width_of(k + 1); c += PT)
            b_base[boff_of(k) + c] = m.small[fw_of(k) + width_of(k) * width_of(k + 1) + c];
        stage_weights<PT>(W_base + woff_of(k), m.small + fw_of(k), width_of(k), width_of(k + 1));
    }
    for (int k = tid; k < TU * dg; k += PT) {
        const int ul = k / (dg > 0 ? dg : 1), c = k % (dg > 0 ? dg : 1);
        ph[k] = u0 + ul < Bt ? m.Pg[(int64_t)users[u0 + ul] * dg + c] * h[c] : 0.f;
    }
    for (int k = tid; k < TU * w1; k += PT)
        Aul[k] = u0 + k / w1 < Bt ? Au[(u0 + k / w1) * w1 + k % w1] : 0.f;

    // ---- tile staging: global -> registers (in flight during the previous tile) -> LDS ------------------------------
    const int nb4 = TI * w1 / 4, nq4 = TI * dg / 4; // float4s per tile
    // row / column of a float4 index: widths are powers of two in practice -> shifts instead of divisions
    const int wb4 = w1 / 4, wq4 = dg / 4;
    const bool p2b = wb4 > 0 && (wb4 & (wb4 - 1)) == 0, p2q = wq4 > 0 && (wq4 & (wq4 - 1)) == 0;
    const int shb = p2b ? __builtin_ctz(wb4) : 0, shq = p2q ? __builtin_ctz(wq4) : 0;
    const auto row_b = [&](int idx) { return p2b ? idx >> shb : idx / (wb4 > 0 ? wb4 : 1); };
    const auto col_b = [&](int idx) { return p2b ? idx & (wb4 - 1) : idx % (wb4 > 0 ? wb4 : 1); };
    const auto row_q = [&](int idx) { return p2q ? idx >> shq : idx / (wq4 > 0 ? wq4 : 1); };
    const auto col_q = [&](int idx) { return p2q ? idx & (wq4 - 1) : idx % (wq4 > 0 ? wq4 : 1); };
    float4 sb[MAXF4], sq[MAXF4];
    const auto stage_load = [&](int64_t it0) {
#pragma unroll
        for (int c = 0; c < MAXF4; ++c) {
            const int idx = tid + c * PT;
            sb[c] = sq[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < nb4) {
                const int r = row_b(idx), k4 = col_b(idx);
                if (it0 + r < I)
                    sb[c] = *reinterpret_cast<const float4 *>(Bi + (it0 + r) * w1 + 4 * k4);
            }
            if (idx < nq4) {
                const int r = row_q(idx), k4 = col_q(idx);
                if (it0 + r < I)
                    sq[c] = *reinterpret_cast<const float4 *>(m.Qg + (it0 + r) * dg + 4 * k4);
            }
        }
    };
    const auto stage_store = [&]() {
#pragma unroll
        for (int c = 0; c < MAXF4; ++c) {
            const int idx = tid + c * PT;
            if (idx < nb4) {
                float *dst = Bl + row_b(idx) * ldb + 4 * col_b(idx);
                dst[0] = sb[c].x, dst[1] = sb[c].y, dst[2] = sb[c].z, dst[3] = sb[c].w;
            }
            if (idx < nq4) {
                float *dst = Ql + row_q(idx) * ldg + 4 * col_q(idx);
                dst[0] = sq[c].x, dst[1] = sq[c].y, dst[2] = sq[c].z, dst[3] = sq[c].w;
            }
        }
    };

    const int64_t n_tiles = (I + TI - 1) / TI;
    int64_t t = blockIdx.x;
    if (t < n_tiles)
        stage_load(t * TI);
    for (; t < n_tiles; t += gridDim.x) {
        const int64_t it0 = t * TI;
        __syncthreads(); // the previous tile's readers are done (first pass: the once-per-workgroup stores are visible)
        stage_store();
        __syncthreads();
        if (t + gridDim.x < n_tiles)
            stage_load((t + gridDim.x) * TI);
        // ---- GMF part: two threads per pair, each over half of the columns ------------------------------------------
        if (has_g) {
            const int p = tid >> 1, half = tid & 1;
            const int ul = p / TI, il = p % TI;
            const int c0 = half * ((dg + 1) / 2), c1 = half ? dg : (dg + 1) / 2;
            float s = 0.f;
#pragma unroll 8
            for (int c = c0; c < c1; ++c)
                s = s + ph[ul * dg + c] * Ql[il * ldg + c];
            s += __shfl_xor(s, 1, 64);
            if (half == 0)
                zg[p] = s;
        } else if (tid < TP) {
            zg[tid] = 0.f;
        }
        // ---- tower layers 1 .. L-1; layer 1 reads relu(Au + Bi) formed on the fly ---------------------------------------
        const float *cur = nullptr;
        for (int k = 1; k < L; ++k) {
            const int kin = width_of(k), kout = width_of(k + 1), lda = kin + 1, ldo = kout + 1, ws = kout + 1;
            const float *Wl = W_base + woff_of(k), *b = b_base + boff_of(k);
            const float *A = cur;
            float *O = (k & 1) ? buf1 : buf2;
            const auto fb = [&](int kk, int c) { return Wl[kk * ws + c]; };
            const auto fs = [&](int p, int c, float s) {
                s += b[c];
                O[p * ldo + c] = s > 0.f ? s : 0.f;
            };
            if (k == 1) // the tall product with the computed A operand: two column blocks per pass
                tile_gemm_wide<NWV>(
                    TP, kout, kin,
                    [&](int p, int kk) {
                        const float v = Aul[(p / TI) * w1 + kk] + Bl[(p % TI) * ldb + kk];
                        return v > 0.f ? v : 0.f;
                    },
                    fb, fs);
            else
                tile_gemm<NWV>(TP, kout, kin, [&](int p, int kk) { return A[p * lda + kk]; }, fb, fs);
            __syncthreads();
            cur = O;
        }
        if (L <= 1)
            __syncthreads(); // zg
        // ---- logit ---------------------------------------------------------------------------------------------------------
        if (tid < TP) {
            const int p = tid, ul = p / TI, il = p % TI;
            float z = zg[p];
            if (L == 1) { // a one-layer tower: its output is relu(Au + Bi) itself
                for (int k = 0; k < last; ++k) {
                    const float v = Aul[ul * w1 + k] + Bl[il * ldb + k];
                    z = z + (v > 0.f ? v : 0.f) * hl[k];
                }
            } else {
                const int ldl = last + 1;
#pragma unroll 8
                for (int k = 0; k < last; ++k)
                    z = z + cur[p * ldl + k] * hl[k];
            }
            const int64_t ur = u0 + ul, it = it0 + il;
            if (ur < Bt && it < I)
                logits[ur * I + it] = z;
        }
    }
}

size_t ncf_pair_lds_floats(bool has_g, bool has_m, int dg, int L0, int n_layers)
{
    const int w1 = has_m ? L0 / 2 : 0, g = has_g ? dg : 0;
    size_t f = (size_t)TI * (w1 + 1) + (size_t)TI * (g + 1) + (size_t)TU * w1 + (size_t)TU * g +
               (size_t)TP * (w1 / 2 + 1) + (size_t)TP * (w1 / 4 + 1) + TP;
    int w = w1;
    for (int k = 1; k < (has_m ? n_layers : 0); ++k) { // b_k and W_k of layers 1..
        f += (size_t)(w / 2) + (size_t)w * (w / 2 + 1);
        w /= 2;
    }
    return f + w; // h of the tower part
}

typedef unsigned long long u64;
__device__ inline uint32_t ordered_bits(float f)
{
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}

__device__ inline float from_ordered(uint32_t o)
{
    const uint32_t u = o ^ ((o >> 31) ? 0x80000000u : 0xffffffffu);
    return __uint_as_float(u);
}

// One workgroup (4 waves) per user row: the row is staged in LDS ([I] floats), listed items struck out (or only they
// kept); every thread keeps the best key (ordered score, ~item: larger is better, ties -> lower item id) among its
// own items c = tid, tid + 256, ...; each of the K rounds is a workgroup arg-max over those 256 keys, after which
// only the winner's owner strikes the item and rescans its own items.
constexpr int TKT = 256;
__global__ __launch_bounds__(TKT) void topk_rows(const float *__restrict__ scores, int64_t I, const int32_t *__restrict__ users,
                                                 const int64_t *__restrict__ seen_ptr,
                                                 const int32_t *__restrict__ seen_items, int K, float sign,
                                                 int32_t *__restrict__ ids, float *__restrict__ vals)
{
    extern __shared__ float row[];
    __shared__ u64 wbest[2][TKT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t r = blockIdx.x;
    const float worst = -INFINITY;
    const bool keep_listed = K >> 16; // the host packs the filter mode above K
    K &= 0xffff;
    const int32_t uu = users[r];
    if (!keep_listed) { // listed = seen in training: struck out
        for (int64_t c = tid; c < I; c += TKT)
            row[c] = sign * scores[r * I + c];
        __syncthreads();
        for (int64_t e = seen_ptr[uu] + tid; e < seen_ptr[uu + 1]; e += TKT)
            row[seen_items[e]] = worst;
    } else { // listed = the user's candidates: only they compete
        for (int64_t c = tid; c < I; c += TKT)
            row[c] = worst;
        __syncthreads();
        for (int64_t e = seen_ptr[uu] + tid; e < seen_ptr[uu + 1]; e += TKT)
            row[seen_items[e]] = sign * scores[r * I + seen_items[e]];
    }
    __syncthreads();
    const auto scan = [&]() {
        u64 b = 0;
        for (int64_t c = tid; c < I; c += TKT) {
            const float s = row[c];
            if (s != worst) {
                const u64 key = ((u64)ordered_bits(s) << 32) | (0xffffffffu - (uint32_t)c);
                b = key > b ? key : b;
            }
        }
        return b;
    };
    u64 mine = scan();
    for (int k = 0; k < K; ++k) {
        u64 best = mine;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const u64 o = __shfl_xor(best, off, 64);
            best = o > best ? o : best;
        }
        if (lane == 0)
            wbest[k & 1][wv] = best;
        __syncthreads(); // (double-buffered by round parity: one barrier per round)
#pragma unroll
        for (int w = 0; w < TKT / 64; ++w)
            best = wbest[k & 1][w] > best ? wbest[k & 1][w] : best;
        if (best == 0) { // nothing left: the rest of the list is padding
            if (tid == 0)
                for (int kk = k; kk < K; ++kk) {
                    ids[r * K + kk] = -1;
                    vals[r * K + kk] = sign * worst;
                }
            return;
        }
        const int item = (int)(0xffffffffu - (uint32_t)best);
        if (tid == 0) {
            ids[r * K + k] = item;
            vals[r * K + k] = sign * from_ordered((uint32_t)(best >> 32));
        }
        if (item % TKT == tid) { // the owner strikes it and finds its next best
            row[item] = worst;
            mine = scan();
        }
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
    }
    lds = sizeof(float) * ncf_pair_lds_floats(d.kind != CREC_NCF_MLP, has_m, d.dg, d.L0, d.n_layers);
    CREC_REQUIRE(ctx, lds <= 160 * 1024, "model too wide for the scoring tile (%zu bytes of LDS)", lds);
    using PairFn = void (*)(NcfDev, const int32_t *, int, int64_t, const float *, const float *, float *);
    PairFn pair_fn = ncf_pair_logits<-1, 0, 0>;
    if (has_m && d.L0 == 128 && d.n_layers == 3 && !ctx->knob_ncf_generic) {
        if (d.kind == CREC_NCF_MLP)
            pair_fn = ncf_pair_logits<128, 3, 0>;
        else if (d.dg == 64)
            pair_fn = ncf_pair_logits<128, 3, 64>;
        else if (d.dg == 32)
            pair_fn = ncf_pair_logits<128, 3, 32>;
    }
    CREC_HIP(ctx, hipFuncSetAttribute((const void *)pair_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CREC_REQUIRE(ctx, d.L0 / 2 <= MAXF4 * 4 * PT / TI && d.dg <= MAXF4 * 4 * PT / TI,
                 "scoring tile: row widths up to %d", MAXF4 * 4 * PT / TI);
    // a workgroup keeps its TU users and walks the item tiles; enough splits of the catalogue for two workgroups per CU's worth of work
    const int64_t user_tiles = crec_ceil_div(Bt, TU), n_tiles = crec_ceil_div(I, TI);
    int64_t splits = crec_ceil_div(2 * (int64_t)(ctx->num_cus > 0 ? ctx->num_cus : 256), user_tiles);
    splits = splits < 1 ? 1 : (splits > n_tiles ? n_tiles : splits);
    const dim3 grid((unsigned)splits, (unsigned)user_tiles);
    crec_prof_begin(ctx, CREC_K_NCF_SCORES, st);
    CREC_KLAUNCH(ctx, pair_fn, grid, dim3(PT), lds, st, d, users, (int)Bt, I, (const float *)Au, (const float *)Bi,
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
    hipLaunchKernelGGL(topk_rows, dim3((unsigned)Bt), dim3(TKT), lds, (hipStream_t)stream, scores, I, users, seen_ptr, seen_items,
                       (int)K | (keep_listed << 16), ascending ? -1.0f : 1.0f, topk_ids, topk_scores);
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}

} // extern "C"
