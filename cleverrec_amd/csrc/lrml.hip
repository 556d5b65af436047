// Latent Relational Metric Learning (LRML) for gfx950: training step and full-catalogue distances.
//
// Replaces, per batch, sess.run([self.train, self.loss], {u_idx, i_idx, j_idx}) (model/RankingRecommender.py:39-47) on the
// graph of model/ranking/LRML.py:36-63 and, for evaluation, sess.run(self.pre_scores) (LRML.py:70-76):
//   joint = p_u * q                      (elementwise, [d])
//   a     = softmax(joint K)             K [d, mem]                                   (LRML.py:49-50)
//   r     = a M                          M [mem, d]: the relation vector              (LRML.py:51)
//   dist  = |p_u + r - q|^2                                                            (LRML.py:60-61)
//   loss  = sum max(dist_i - dist_j + margin, 0) + reg * (l2(p_u) + l2(q_i) + l2(q_j)) (LRML.py:63, utils/tools.py:72-73)
// with tf.train.AdamOptimizer: P and Q are reached through tf.gather (IndexedSlices: duplicate rows summed, then every
// row's m, v decayed and var moved), K and M are dense variables (ApplyAdam).  _unit_clipping is defined but
// never called (LRML.py:65-68, :78-84), and the variables' `regularizer=` never reaches the loss.
//
// Per batch, three launches (like the NCF family, csrc/ncf.hip):
//   lrml_tile       one workgroup (8 waves) per tile of 16 triples = 32 (user, item) pairs, everything on chip: K, M, the
//                   gathered rows, joint, attention and relation vectors live in LDS; the four forward / backward products
//                   and the two parameter-gradient products are 16x16x4 fp32 MFMA tiles (tile_gemm, ncf_common.h).
//                   Writes contribution rows for P and Q (without the l2 term), the item slot ids, per-triple losses and
//                   the tile's partial dK | dM.
//   rows_index      the batch's 2B item slots sorted by row (rows_adam.hip); the user index is built once per epoch
//   crec_apply_all  P, Q (TF1 sparse Adam over every row) and [K | M] (tile partials summed in fixed slices, dense Adam).
// Deterministic: fixed summation orders, no atomics.
#include "crec_internal.h"
#include "crec_store.h"
#include "rows_adam.h"
#include "ncf_common.h"

namespace {

constexpr int T = 32;      // pairs per tile: rows 0..15 the (u, i) pairs of 16 triples, rows 16..31 their (u, j) pairs
constexpr int TT = T / 2;  // triples per tile
constexpr int NT = 512;    // threads per workgroup
constexpr int NW = NT / 64;
constexpr int LPI = NT / T; // lanes per pair in the row-wise phases (16)

__device__ __forceinline__ float grp16_sum(float v)
{
#pragma unroll
    for (int off = 1; off < LPI; off <<= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float grp16_max(float v)
{
#pragma unroll
    for (int off = 1; off < LPI; off <<= 1) {
        const float o = __shfl_xor(v, off, 64);
        v = o > v ? o : v;
    }
    return v;
}

// K [d][mem] and M [mem][d] from the flat block [K | M] into LDS, rows padded by one float; a wave per row
__device__ __forceinline__ void stage_km(float *__restrict__ Kl, float *__restrict__ Ml, const float *__restrict__ small, int d, int mem,
                                         int wave, int lane)
{
    const int lm = mem + 1, ld = d + 1;
    for (int r = wave; r < d; r += NW)
        for (int c = lane; c < mem; c += 64)
            Kl[r * lm + c] = small[(int64_t)r * mem + c];
    const float *Mg = small + (int64_t)d * mem;
    for (int r = wave; r < mem; r += NW)
        for (int c = lane; c < d; c += 64)
            Ml[r * ld + c] = Mg[(int64_t)r * d + c];
}

// softmax over the first `mem` columns of each of the tile's T rows, in place (tf.nn.softmax: exp(x - max) / sum)
__device__ __forceinline__ void softmax_rows(float *__restrict__ At, int mem)
{
    const int r = threadIdx.x / LPI, sub = threadIdx.x % LPI, lm = mem + 1;
    float mx = -3.4e38f;
    for (int s = sub; s < mem; s += LPI)
        mx = fmaxf(mx, At[r * lm + s]);
    mx = grp16_max(mx);
    float sum = 0.f;
    for (int s = sub; s < mem; s += LPI) {
        const float e = expf(At[r * lm + s] - mx);
        At[r * lm + s] = e;
        sum = sum + e;
    }
    sum = grp16_sum(sum);
    for (int s = sub; s < mem; s += LPI)
        At[r * lm + s] = At[r * lm + s] / sum;
}

size_t lrml_tile_lds_floats(int d, int mem) { return (size_t)(TT + 4 * T) * (d + 1) + 2 * (size_t)T * (mem + 1) + (size_t)d * (mem + 1) + (size_t)mem * (d + 1) + 4 * T; }

// <DC, MC>: embed_size and mem_size fixed at compile time (the shipped conf's 128 x 50), or 0 = the arguments; with the
// shape fixed the products' block loops, LDS offsets and edge handling fold away (as for the NCF tile, csrc/ncf.hip)
template <int DC, int MC>
__global__ __launch_bounds__(NT) void lrml_tile(const float *__restrict__ P, const float *__restrict__ Q, const float *__restrict__ small,
                                                int d_arg, int mem_arg, const int32_t *__restrict__ u, const int32_t *__restrict__ i,
                                                const int32_t *__restrict__ j, int nb, float reg, float margin,
                                                float *__restrict__ c_p, float *__restrict__ c_q, int32_t *__restrict__ slots,
                                                float *__restrict__ loss_inst, float *__restrict__ partial)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int d = DC > 0 ? DC : d_arg, mem = MC > 0 ? MC : mem_arg;
    const int ld = d + 1, lm = mem + 1, n_small = 2 * d * mem;
    float *const Pu = lds;            // [TT][ld] user rows
    float *const Qx = Pu + TT * ld;   // [T][ld]  item rows (i then j)
    float *const J = Qx + T * ld;     // [T][ld]  joint = p_u * q
    float *const R = J + T * ld;      // [T][ld]  relation vector, then diff = p_u + r - q
    float *const DJ = R + T * ld;     // [T][ld]  dL/djoint
    float *const At = DJ + T * ld;    // [T][lm]  logits, then attention
    float *const Dl = At + T * lm;    // [T][lm]  dL/dattention, then dL/dlogits
    float *const Kl = Dl + T * lm;    // [d][lm]
    float *const Ml = Kl + d * lm;    // [mem][ld]
    float *const ev = Ml + mem * ld;  // [T] dL/ddist * 2 per pair (0 for pairs past the batch's end)
    float *const ds = ev + T;         // [T] dist per pair
    int *const su = (int *)(ds + T);  // [TT] user ids, then [T] item ids
    int *const si = su + TT;

    const int tile = blockIdx.x, t0 = tile * TT;
    const int nr = (nb - t0) < TT ? (nb - t0) : TT; // triples of this tile
    if (tid < TT) {
        const int t = t0 + (tid < nr ? tid : nr - 1); // rows past the end repeat the last triple; their e is 0
        const int uu = u[t], ii = i[t], jj = j[t];
        su[tid] = uu;
        si[tid] = ii;
        si[TT + tid] = jj;
        if (tid < nr) { // item slots of the batch: [i-slots | j-slots], nb each
            slots[t] = ii;
            slots[nb + t] = jj;
        }
    }
    stage_km(Kl, Ml, small, d, mem, wave, lane);
    __syncthreads();
    // gathers: lanes along the row, two user rows and four item rows per wave
    for (int rr = 0; rr < TT / NW; ++rr) {
        const int r = wave * (TT / NW) + rr;
        const float *src = P + (int64_t)su[r] * d;
        for (int c = lane; c < d; c += 64)
            Pu[r * ld + c] = src[c];
    }
    for (int rr = 0; rr < T / NW; ++rr) {
        const int r = wave * (T / NW) + rr;
        const float *src = Q + (int64_t)si[r] * d;
        for (int c = lane; c < d; c += 64)
            Qx[r * ld + c] = src[c];
    }
    __syncthreads();
    for (int rr = 0; rr < T / NW; ++rr) {
        const int r = wave * (T / NW) + rr;
        for (int c = lane; c < d; c += 64)
            J[r * ld + c] = Pu[(r & (TT - 1)) * ld + c] * Qx[r * ld + c];
    }
    __syncthreads();
    // ---- forward: logits = joint K, a = softmax, r = a M, diff = (p_u + r) - q ------------------------------------
    tile_gemm<NW>(
        T, mem, d, [&](int r, int k) { return J[r * ld + k]; }, [&](int k, int s) { return Kl[k * lm + s]; },
        [&](int r, int s, float v) { At[r * lm + s] = v; });
    __syncthreads();
    softmax_rows(At, mem);
    __syncthreads();
    tile_gemm<NW>(
        T, d, mem, [&](int r, int s) { return At[r * lm + s]; }, [&](int s, int c) { return Ml[s * ld + c]; },
        [&](int r, int c, float v) { R[r * ld + c] = (Pu[(r & (TT - 1)) * ld + c] + v) - Qx[r * ld + c]; });
    __syncthreads();
    // ---- distances, hinge, loss ------------------------------------------------------------------------------------
    {
        const int r = tid / LPI, sub = tid % LPI;
        float dd = 0.f, nq = 0.f, np = 0.f;
        for (int c = sub; c < d; c += LPI) {
            const float x = R[r * ld + c], q = Qx[r * ld + c], p = Pu[(r & (TT - 1)) * ld + c];
            dd = dd + x * x;
            nq = nq + q * q;
            np = np + p * p;
        }
        dd = grp16_sum(dd);
        nq = grp16_sum(nq);
        np = grp16_sum(np);
        if (sub == 0) {
            ds[r] = dd;
            ev[r] = nq;          // parked: the hinge below replaces it
            Dl[r * lm] = np;     // (Dl is free until the backward pass)
        }
    }
    __syncthreads();
    if (tid < TT) {
        const float x = ds[tid] - ds[TT + tid], hm = x + margin;
        const bool ok = tid < nr;
        const float g = ok && hm >= 0.f ? 2.0f : 0.f; // d max(y + margin, 0) / dy = 1 where y + margin >= 0 (tf.maximum), times d|.|^2
        const float nqi = ev[tid], nqj = ev[TT + tid], np = Dl[tid * lm];
        if (ok) // tf.nn.l2_loss = sum(x^2) / 2 of each gathered row (LRML.py:63)
            loss_inst[t0 + tid] = (hm > 0.f ? hm : 0.f) + reg * ((0.5f * np + 0.5f * nqi) + 0.5f * nqj);
        ev[tid] = g;
        ev[TT + tid] = -g;
    }
    __syncthreads();
    // ---- backward: dR = e * diff; dA = dR M^T; dlogits = a * (dA - <a, dA>); dJ = dlogits K^T ------------------------
    tile_gemm<NW>(
        T, mem, d, [&](int r, int c) { return ev[r] * R[r * ld + c]; }, [&](int c, int s) { return Ml[s * ld + c]; },
        [&](int r, int s, float v) { Dl[r * lm + s] = v; });
    __syncthreads();
    {
        const int r = tid / LPI, sub = tid % LPI;
        float dot = 0.f;
        for (int s = sub; s < mem; s += LPI)
            dot = dot + At[r * lm + s] * Dl[r * lm + s];
        dot = grp16_sum(dot);
        for (int s = sub; s < mem; s += LPI)
            Dl[r * lm + s] = At[r * lm + s] * (Dl[r * lm + s] - dot);
    }
    __syncthreads();
    float *part = partial + (int64_t)tile * n_small;
    tile_gemm<NW>(
        T, d, mem, [&](int r, int s) { return Dl[r * lm + s]; }, [&](int s, int c) { return Kl[c * lm + s]; },
        [&](int r, int c, float v) { DJ[r * ld + c] = v; });
    // dK[c][s] = sum_r joint[r][c] * dlogits[r][s];  dM[s][c] = sum_r a[r][s] * dR[r][c]   (this tile's share)
    tile_gemm<NW>(
        d, mem, T, [&](int c, int r) { return J[r * ld + c]; }, [&](int r, int s) { return Dl[r * lm + s]; },
        [&](int c, int s, float v) { st_out(part + c * mem + s, v); });
    tile_gemm<NW>(
        mem, d, T, [&](int s, int r) { return At[r * lm + s]; }, [&](int r, int c) { return ev[r] * R[r * ld + c]; },
        [&](int s, int c, float v) { st_out(part + d * mem + s * d + c, v); });
    __syncthreads();
    // ---- contribution rows (without the l2 term: the update adds reg * row per occurrence) --------------------------
    //   d/dq   = dJ * p_u - dR          d/dp_u = (dJ_i * q_i + dR_i) + (dJ_j * q_j + dR_j)
    for (int rr = 0; rr < T / NW; ++rr) {
        const int r = wave * (T / NW) + rr, t = r & (TT - 1);
        if (t < nr) {
            float *dst = c_q + (int64_t)((r < TT ? 0 : nb) + t0 + t) * d;
            for (int c = lane; c < d; c += 64)
                st_out(dst + c, DJ[r * ld + c] * Pu[t * ld + c] - ev[r] * R[r * ld + c]);
        }
    }
    for (int rr = 0; rr < TT / NW; ++rr) {
        const int t = wave * (TT / NW) + rr;
        if (t < nr) {
            float *dst = c_p + (int64_t)(t0 + t) * d;
            for (int c = lane; c < d; c += 64)
                st_out(dst + c, (DJ[t * ld + c] * Qx[t * ld + c] + ev[t] * R[t * ld + c]) +
                                    (DJ[(TT + t) * ld + c] * Qx[(TT + t) * ld + c] + ev[TT + t] * R[(TT + t) * ld + c]));
        }
    }
}

// Full-catalogue distances of a block of users: scores[b][item] = |p_u + r(u, item) - q_item|^2 (LRML.py:70-76).
// A workgroup owns one user and walks item tiles of T rows (blockIdx.x, stride gridDim.x); K, M and p_u stay in LDS.
size_t lrml_scores_lds_floats(int d, int mem) { return (size_t)(1 + 2 * T) * (d + 1) + (size_t)T * (mem + 1) + (size_t)d * (mem + 1) + (size_t)mem * (d + 1); }

template <int DC, int MC>
__global__ __launch_bounds__(NT) void lrml_scores(const float *__restrict__ P, const float *__restrict__ Q, const float *__restrict__ small,
                                                  int d_arg, int mem_arg, const int32_t *__restrict__ users, int64_t I, float *__restrict__ scores)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int d = DC > 0 ? DC : d_arg, mem = MC > 0 ? MC : mem_arg;
    const int ld = d + 1, lm = mem + 1;
    float *const pu = lds;           // [ld]
    float *const Qx = pu + ld;       // [T][ld]
    float *const R = Qx + T * ld;    // [T][ld]
    float *const At = R + T * ld;    // [T][lm]
    float *const Kl = At + T * lm;   // [d][lm]
    float *const Ml = Kl + d * lm;   // [mem][ld]
    const int64_t b = blockIdx.y;
    const float *prow = P + (int64_t)users[b] * d;
    for (int c = tid; c < d; c += NT)
        pu[c] = prow[c];
    stage_km(Kl, Ml, small, d, mem, wave, lane);
    const int64_t n_tiles = (I + T - 1) / T;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t i0 = tile * T;
        const int nr = (int)((I - i0) < T ? (I - i0) : T);
        __syncthreads(); // the previous tile's readers are done (first pass: p_u, K, M staged)
        for (int rr = 0; rr < T / NW; ++rr) {
            const int r = wave * (T / NW) + rr;
            const float *src = Q + (i0 + (r < nr ? r : nr - 1)) * d;
            for (int c = lane; c < d; c += 64)
                Qx[r * ld + c] = src[c];
        }
        __syncthreads();
        tile_gemm<NW>(
            T, mem, d, [&](int r, int k) { return pu[k] * Qx[r * ld + k]; }, [&](int k, int s) { return Kl[k * lm + s]; },
            [&](int r, int s, float v) { At[r * lm + s] = v; });
        __syncthreads();
        softmax_rows(At, mem);
        __syncthreads();
        tile_gemm<NW>(
            T, d, mem, [&](int r, int s) { return At[r * lm + s]; }, [&](int s, int c) { return Ml[s * ld + c]; },
            [&](int r, int c, float v) { R[r * ld + c] = (pu[c] + v) - Qx[r * ld + c]; });
        __syncthreads();
        const int r = tid / LPI, sub = tid % LPI;
        float dd = 0.f;
        for (int c = sub; c < d; c += LPI)
            dd = dd + R[r * ld + c] * R[r * ld + c];
        dd = grp16_sum(dd);
        if (sub == 0 && r < nr)
            scores[b * I + i0 + r] = dd;
    }
}

__global__ __launch_bounds__(256) void lrml_reduce_loss(const float *__restrict__ loss_inst, int64_t n_total, int32_t B,
                                                        float *__restrict__ loss_per_batch)
{
    __shared__ double part[256];
    const int64_t t0 = (int64_t)blockIdx.x * B;
    const int nb = (int)((n_total - t0) < B ? (n_total - t0) : B);
    double s = 0.0;
    for (int r = threadIdx.x; r < nb; r += 256)
        s += (double)loss_inst[t0 + r];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        loss_per_batch[blockIdx.x] = (float)part[0];
}

int lrml_check(crec_ctx *ctx, const crec_lrml *m, int *d_out, int *mem_out)
{
    CREC_REQUIRE(ctx, m && m->P.var && m->Q.var && m->small.var && m->P.cols == m->Q.cols, "LRML variables: P [U,d], Q [I,d], [K | M]");
    const int d = m->P.cols, mem = m->mem_size;
    CREC_REQUIRE(ctx, d % 4 == 0 && d >= 4 && d <= 128, "embed_size %d: need a multiple of 4 in [4, 128] (the tile keeps K, M and 32 pairs in LDS)", d);
    CREC_REQUIRE(ctx, mem >= 1 && mem <= 64, "mem_size %d: need 1..64", mem);
    CREC_REQUIRE(ctx, m->small.rows * m->small.cols == 2 * (int64_t)d * mem, "the [K | M] block must hold 2 * embed_size * mem_size floats");
    *d_out = d;
    *mem_out = mem;
    return CREC_OK;
}

} // namespace

extern "C" int crec_lrml_train_epoch(crec_ctx *ctx, const crec_lrml *m, crec_adam *opt, const int32_t *u, const int32_t *i,
                                     const int32_t *j, int64_t N, int32_t B, float *loss_per_batch, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, m && opt, "null argument");
    int d = 0, mem = 0;
    if (int rc = lrml_check(ctx, m, &d, &mem))
        return rc;
    CREC_REQUIRE(ctx, m->small.m && m->small.v && m->P.m && m->P.v && m->Q.m && m->Q.v, "Adam slots");
    CREC_REQUIRE(ctx, B >= 1 && B <= 65536, "batch_size <= 65536"); // (user slots indexed 16 384 at a time, item slots 2048 at a time in <= 64 chunks)
    const int u_chunks = (int)crec_ceil_div(B, 16384);
    CREC_REQUIRE(ctx, N >= 0 && (N == 0 || (u && i && j && loss_per_batch)), "null argument");
    if (N == 0)
        return CREC_OK;
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t U = m->P.rows, I = m->Q.rows, nb = crec_ceil_div(N, B);
    const int n_small = 2 * d * mem, max_tiles = (int)crec_ceil_div(B, TT);
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const int q_chunks = (int)crec_ceil_div(2 * (int64_t)B, 2048); // item slots: sorted 2048 at a time
    const size_t ixu = crec_rows_index_bytes(nb, B, U, u_chunks), ixq = crec_rows_index_bytes(1, 2 * B, I, q_chunks);
    const size_t cpb = sizeof(float) * (size_t)B * d, cqb = sizeof(float) * (size_t)2 * B * d;
    const size_t sl = sizeof(int32_t) * (size_t)2 * B, li = sizeof(float) * (size_t)N, pb = sizeof(float) * (size_t)max_tiles * n_small;
    if (int rc = crec_reserve(ctx, &ctx->idx_ws, up(ixu) + up(ixq) + up(cpb) + up(cqb) + up(sl) + up(li) + up(pb)))
        return rc;
    char *ws = (char *)ctx->idx_ws.ptr;
    crec_rows_index_view vu, vq;
    if (int rc = crec_rows_index(ctx, u, N, B, U, ws, &vu, st, u_chunks))
        return rc;
    char *p = ws + up(ixu);
    char *ixq_ws = p;
    p += up(ixq);
    float *c_p = (float *)p;
    p += up(cpb);
    float *c_q = (float *)p;
    p += up(cqb);
    int32_t *slots = (int32_t *)p;
    p += up(sl);
    float *loss_inst = (float *)p;
    p += up(li);
    float *partial = (float *)p;
    const size_t lds = lrml_tile_lds_floats(d, mem) * sizeof(float);
    CREC_REQUIRE(ctx, lds <= 160 * 1024, "LRML tile: %zu bytes of LDS", lds);
    const bool fixed = d == 128 && mem == 50 && !ctx->knob_ncf_generic; // the shipped conf (conf/LRML.properties)
    const auto tile_fn = fixed ? lrml_tile<128, 50> : lrml_tile<0, 0>;
    CREC_HIP(ctx, hipFuncSetAttribute((const void *)tile_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t t0 = b * B;
        const int nbt = (int)((N - t0) < B ? (N - t0) : B);
        const int tiles = (int)crec_ceil_div(nbt, TT);
        crec_rows_opt o{};
        o.kind = ROWS_SPARSE_TF1;
        o.reg_occ = m->reg;
        if (int rc = crec_rows_optimizer(ctx, opt, &o, true))
            return rc;
        hipLaunchKernelGGL(tile_fn, dim3((unsigned)tiles), dim3(NT), lds, st, (const float *)m->P.var, (const float *)m->Q.var,
                           (const float *)m->small.var, d, mem, u + t0, i + t0, j + t0, nbt, m->reg, m->margin, c_p, c_q, slots,
                           loss_inst + t0, partial);
        // the batch's item slots are laid out [i | j] with stride nbt (not B), so a ragged batch has no holes
        if (int rc = crec_rows_index(ctx, slots, 2 * (int64_t)nbt, 2 * B, I, ixq_ws, &vq, st, q_chunks))
            return rc;
        crec_apply_set set{}; // (the item table first: its popular rows' waves are the launch's longest)
        set.tables[set.n_tables++] = crec_rows_job{crec_rows_table{m->Q.var, m->Q.m, m->Q.v, I, d}, vq.chunks, vq.rowseg, vq.order, c_q,
                                                   nullptr, nullptr, o};
        set.tables[set.n_tables++] = crec_rows_job{crec_rows_table{m->P.var, m->P.m, m->P.v, U, d}, vu.chunks,
                                                   vu.rowseg + b * vu.chunks * vu.rows, vu.order + b * vu.per_batch, c_p, nullptr, nullptr, o};
        crec_rows_opt od = o; // K, M: dense variables, plain ApplyAdam on the summed tile partials
        od.kind = ROWS_DENSE;
        od.reg_occ = 0.f;
        set.small = crec_small_job{m->small.var, m->small.m, m->small.v, partial, tiles, n_small, n_small, od};
        if (int rc = crec_apply_all(ctx, set, st))
            return rc;
        opt->beta1_power = opt->beta1_power * opt->beta1;
        opt->beta2_power = opt->beta2_power * opt->beta2;
        opt->step += 1;
    }
    hipLaunchKernelGGL(lrml_reduce_loss, dim3((unsigned)nb), dim3(256), 0, st, (const float *)loss_inst, N, B, loss_per_batch);
    CREC_HIP(ctx, hipGetLastError());
    return crec_prof_flush(ctx, st);
}

extern "C" int crec_lrml_scores(crec_ctx *ctx, const crec_lrml *m, const int32_t *users, int32_t Bt, float *scores, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    int d = 0, mem = 0;
    if (int rc = lrml_check(ctx, m, &d, &mem))
        return rc;
    CREC_REQUIRE(ctx, Bt >= 0 && Bt <= 65535 && (Bt == 0 || (users && scores)), "users / scores");
    if (Bt == 0)
        return CREC_OK;
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t I = m->Q.rows;
    const size_t lds = lrml_scores_lds_floats(d, mem) * sizeof(float);
    CREC_REQUIRE(ctx, lds <= 160 * 1024, "LRML scores: %zu bytes of LDS", lds);
    const bool fixed = d == 128 && mem == 50 && !ctx->knob_ncf_generic;
    const auto sc_fn = fixed ? lrml_scores<128, 50> : lrml_scores<0, 0>;
    CREC_HIP(ctx, hipFuncSetAttribute((const void *)sc_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // enough workgroups per user to fill the chip even for a handful of users, few enough that staging K, M is amortised
    const int64_t n_tiles = crec_ceil_div(I, T);
    int per_user = (int)crec_ceil_div(1024, Bt);
    per_user = per_user < 1 ? 1 : (per_user > 16 ? 16 : per_user);
    if (per_user > n_tiles)
        per_user = (int)n_tiles;
    hipLaunchKernelGGL(sc_fn, dim3((unsigned)per_user, (unsigned)Bt), dim3(NT), lds, st, (const float *)m->P.var,
                       (const float *)m->Q.var, (const float *)m->small.var, d, mem, users, I, scores);
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}
