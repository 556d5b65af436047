// Shared by ncf.hip (training) and ncf_eval.hip (full-catalogue scoring): the device-side view of a crec_ncf.
#pragma once
#include "crec_internal.h"

constexpr int MAX_LAYERS = 4;

struct NcfDev {
    int kind; // 0 GMF, 1 MLP, 2 NeuMF
    const float *Pg, *Qg, *Pm, *Qm;
    const float *small; // [W_0 | b_0 | W_1 | b_1 | ... | h]
    int dg, n_layers, L0;
    int w_off[MAX_LAYERS], b_off[MAX_LAYERS], h_off, n_small;
    float reg_g, reg_m;
    int loss; // CREC_LOSS_*: sigmoid cross-entropy | square (utils/tools.py:66-76)
    long long *dbg; // -DNCF_STAMPS builds only: per-phase clock stamps of one tile (tools/ncf_variants.sh)
};

// validates a crec_ncf and lays out the flat small-parameter block [W_0 | b_0 | ... | h]
inline int ncf_describe(crec_ctx *ctx, const crec_ncf *mdl, NcfDev *d)
{
    CREC_REQUIRE(ctx, mdl->kind >= CREC_NCF_GMF && mdl->kind <= CREC_NCF_NEUMF, "ncf kind");
    const bool has_g = mdl->kind != CREC_NCF_MLP, has_m = mdl->kind != CREC_NCF_GMF;
    *d = NcfDev{};
    d->kind = mdl->kind;
    d->reg_g = mdl->reg_gmf;
    d->reg_m = mdl->reg_mlp;
    CREC_REQUIRE(ctx, mdl->loss_func == CREC_LOSS_CROSS_ENTROPY || mdl->loss_func == CREC_LOSS_SQUARE, "loss_func");
    d->loss = mdl->loss_func;
    int off = 0;
    if (has_g) {
        CREC_REQUIRE(ctx, mdl->P_gmf.var && mdl->Q_gmf.var && mdl->P_gmf.cols == mdl->Q_gmf.cols, "gmf tables");
        d->Pg = mdl->P_gmf.var;
        d->Qg = mdl->Q_gmf.var;
        d->dg = mdl->P_gmf.cols;
        CREC_REQUIRE(ctx, d->dg % 4 == 0 && d->dg >= 4 && d->dg <= 256, "embed_size %d: need a multiple of 4 in [4, 256]", d->dg);
    }
    if (has_m) {
        CREC_REQUIRE(ctx, mdl->P_mlp.var && mdl->Q_mlp.var && mdl->P_mlp.cols == mdl->Q_mlp.cols, "mlp tables");
        CREC_REQUIRE(ctx, mdl->n_layers >= 1 && mdl->n_layers <= MAX_LAYERS, "1..4 MLP layers");
        d->Pm = mdl->P_mlp.var;
        d->Qm = mdl->Q_mlp.var;
        d->n_layers = mdl->n_layers;
        d->L0 = 2 * mdl->P_mlp.cols;
        CREC_REQUIRE(ctx, mdl->P_mlp.cols % 4 == 0 && d->L0 <= 512, "layers[0] %d: need a multiple of 8, at most 512", d->L0);
        int w = d->L0;
        for (int k = 0; k < d->n_layers; ++k) { // W_k [w, w/2], b_k [w/2]  (MLP.py:34-37)
            CREC_REQUIRE(ctx, w % 2 == 0 && w >= 2, "layer widths must halve");
            d->w_off[k] = off;
            off += w * (w / 2);
            d->b_off[k] = off;
            off += w / 2;
            w /= 2;
        }
        d->h_off = off;
        off += d->dg + w;
    } else {
        d->h_off = 0;
        off = d->dg;
    }
    d->n_small = off;
    CREC_REQUIRE(ctx, mdl->small.var && mdl->small.m && mdl->small.v && mdl->small.rows * mdl->small.cols == off,
                 "small parameter block must hold %d floats ([W_0|b_0|...|h])", off);
    d->small = mdl->small.var;
    return CREC_OK;
}

// ---- device helpers shared by the training tile (ncf.hip) and the scoring tile (ncf_eval.hip) -----------------
#ifdef __HIPCC__
#ifndef NCF_ABL
#define NCF_ABL 0
#endif
using f32x4 = __attribute__((ext_vector_type(4))) float;

// C[M x N] = A[M x K] * B[K x N] on v_mfma_f32_16x16x4_f32: 16x16 output blocks dealt round-robin to the waves.
// fa(m, k) / fb(k, n) read the operands (LDS), fs(m, n, v) consumes an output; the edges of M / N / K are handled
// here (fa / fb only ever see valid indices).
// Lane l feeds A[l & 15][k_l] and B[k_l][l & 15] and the instruction sums over the four k_l of the quarter-waves
// q = l >> 4; output register j of lane l is C[4 * q + j][l & 15].  Which four k an instruction covers is free as long
// as A and B agree, so K is walked in chunks of 4*S columns with k_l = chunk + s + S*q in step s: with S = 16 and row
// strides that are odd, a half-wave's 32 operand reads fall in 32 different LDS banks whether k runs along the row
// (A of the forward pass) or down the column (B).  The order of the sum is fixed: chunks ascending, s ascending, q
// ascending inside the instruction.
// Two accumulator chains (even / odd steps of a chunk, added once at the end): a dependent MFMA waits for its
// predecessor's result, and a wave of the training tile has one block at a time -- nothing else to interleave with.
template <int S, class FA, class FB>
__device__ __forceinline__ void gemm_chunk(f32x4 &acc0, f32x4 &acc1, int k0, int mc, int nc, int q, FA fa, FB fb)
{
    float a[S], b[S];
#pragma unroll
    for (int s = 0; s < S; ++s) { // all 2*S operand reads in flight together
#if NCF_ABL == 2 // timing ablation (wrong results): no operand reads
        a[s] = (float)(mc + s), b[s] = (float)(nc + k0);
#else
        a[s] = fa(mc, k0 + s + S * q);
        b[s] = fb(k0 + s + S * q, nc);
#endif
    }
    __builtin_amdgcn_sched_barrier(0); // keep the reads batched: the scheduler would otherwise pair each with its MFMA
#pragma unroll
    for (int s = 0; s < S; s += 2) {
#if NCF_ABL == 1 // timing ablation (wrong results): no MFMA
        acc0[0] += a[s] * b[s], acc1[0] += a[s + 1] * b[s + 1];
#else
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s + 1], b[s + 1], acc1, 0, 0, 0);
#endif
    }
}

template <int NWV, class FA, class FB, class FS>
__device__ __forceinline__ void tile_gemm(int M, int N, int K, FA fa, FB fb, FS fs)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, l16 = lane & 15, q = lane >> 4;
    const int nbn = (N + 15) >> 4, nblk = ((M + 15) >> 4) * nbn;
    for (int blk = wave; blk < nblk; blk += NWV) {
        const int m0 = (blk / nbn) << 4, n0 = (blk % nbn) << 4;
        // rows / columns past the edge read a clamped (valid) address: they only feed outputs that are dropped
        // below, so no predicate -- a predicated LDS read costs a branch each
        const int ni = n0 + l16;
        const int mc = (m0 + l16) < M ? (m0 + l16) : M - 1, nc = ni < N ? ni : N - 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 64 <= K; k += 64)
            gemm_chunk<16>(acc, acc1, k, mc, nc, q, fa, fb);
        if (k + 32 <= K) {
            gemm_chunk<8>(acc, acc1, k, mc, nc, q, fa, fb);
            k += 32;
        }
        if (k + 16 <= K) {
            gemm_chunk<4>(acc, acc1, k, mc, nc, q, fa, fb);
            k += 16;
        }
        for (; k < K; k += 4) { // K tail: columns past the end contribute zeros
            const bool kok = k + q < K;
            const int kc = kok ? k + q : K - 1;
            const float a = fa(mc, kc), b = fb(kc, nc);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kok ? a : 0.f, b, acc, 0, 0, 0);
        }
        if (ni < N) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (m0 + q * 4 + j < M)
                    fs(m0 + q * 4 + j, ni, acc[j] + acc1[j]);
        }
    }
}

// sum over the 16 lanes of a DPP row, the total in every lane: xor 1 and xor 2 inside the quads, then the mirrors of 8 and of 16
// lanes -- four VALU instructions where four __shfl_xor are four LDS round trips
template <int CTRL> __device__ __forceinline__ float dpp_f32(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v)
{
    v = v + dpp_f32<0xB1>(v);  // quad_perm [1, 0, 3, 2]
    v = v + dpp_f32<0x4E>(v);  // quad_perm [2, 3, 0, 1]
    v = v + dpp_f32<0x141>(v); // row_half_mirror
    v = v + dpp_f32<0x140>(v); // row_mirror
    return v;
}

// The same product with each wave taking a row block and TWO neighbouring column blocks at a time (N is walked in
// steps of 32): the A operand -- which may be computed on the fly by fa -- is read once for both, and the two
// accumulator chains interleave, so a dependent MFMA never waits on its predecessor.  For tall products (many row
// blocks per wave); with few row blocks tile_gemm keeps more waves busy.
template <int S, class FA, class FB>
__device__ __forceinline__ void gemm_chunk2(f32x4 &acc0, f32x4 &acc1, int k0, int mc, int nc0, int nc1, int q, FA fa, FB fb)
{
    float a[S], b0[S], b1[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        a[s] = fa(mc, k0 + s + S * q);
        b0[s] = fb(k0 + s + S * q, nc0);
        b1[s] = fb(k0 + s + S * q, nc1);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[s], acc1, 0, 0, 0);
    }
}

template <int NWV, class FA, class FB, class FS>
__device__ __forceinline__ void tile_gemm_wide(int M, int N, int K, FA fa, FB fb, FS fs)
{
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, l16 = lane & 15, q = lane >> 4;
    const int nbn = (N + 31) >> 5, nblk = ((M + 15) >> 4) * nbn;
    for (int blk = wave; blk < nblk; blk += NWV) {
        const int m0 = (blk / nbn) << 4, n0 = (blk % nbn) << 5;
        const int ni0 = n0 + l16, ni1 = n0 + 16 + l16;
        const int mc = (m0 + l16) < M ? (m0 + l16) : M - 1, nc0 = ni0 < N ? ni0 : N - 1, nc1 = ni1 < N ? ni1 : N - 1;
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 64 <= K; k += 64)
            gemm_chunk2<16>(acc0, acc1, k, mc, nc0, nc1, q, fa, fb);
        if (k + 32 <= K) {
            gemm_chunk2<8>(acc0, acc1, k, mc, nc0, nc1, q, fa, fb);
            k += 32;
        }
        if (k + 16 <= K) {
            gemm_chunk2<4>(acc0, acc1, k, mc, nc0, nc1, q, fa, fb);
            k += 16;
        }
        for (; k < K; k += 4) { // K tail: columns past the end contribute zeros
            const bool kok = k + q < K;
            const int kc = kok ? k + q : K - 1;
            const float a = kok ? fa(mc, kc) : 0.f;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, fb(kc, nc0), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, fb(kc, nc1), acc1, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (m0 + q * 4 + j < M) {
                if (ni0 < N)
                    fs(m0 + q * 4 + j, ni0, acc0[j]);
                if (ni1 < N)
                    fs(m0 + q * 4 + j, ni1, acc1[j]);
            }
    }
}

// stage a row-major [rows x cols] weight matrix in LDS with row stride cols + 1
template <int NTH>
__device__ __forceinline__ void stage_weights(float *__restrict__ Wl, const float *__restrict__ W, int rows, int cols)
{
    const int n = rows * cols, ws = cols + 1;
    if ((cols & (cols - 1)) == 0 && cols >= 4 && ((uintptr_t)W & 15) == 0) { // the usual case: no division, 16 B loads
        const int sh = __builtin_ctz(cols);
        const float4 *W4 = (const float4 *)W;
#pragma unroll 8
        for (int o4 = threadIdx.x; o4 < n / 4; o4 += NTH) {
            const float4 w = W4[o4];
            float *dst = Wl + ((o4 * 4) >> sh) * ws + ((o4 * 4) & (cols - 1));
            dst[0] = w.x, dst[1] = w.y, dst[2] = w.z, dst[3] = w.w;
        }
    } else {
#pragma unroll 4
        for (int o = threadIdx.x; o < n; o += NTH)
            Wl[(o / cols) * ws + (o % cols)] = W[o];
    }
}
#endif
