// Pointwise NCF family for gfx950: GMF, MLP and NeuMF training steps (sigmoid cross-entropy on
// (u, i, y) instances from pointwise_ranking_sampler).
//
// Replaces sess.run([train, loss], {u_idx, i_idx, y}) (model/RankingRecommender.py:52-58) on the graphs
// of model/ranking/GMF.py:29-57, MLP.py:31-77, NeuMF.py:33-105; see include/crec.h (crec_ncf).
//
// Per batch:
//   ncf_tile        one workgroup per tile of T = 32 instances: gathers the embedding rows into LDS,
//                   GMF product, the ReLU tower forward, logit, e = sigmoid(logit) - y, loss term,
//                   tower backward -- all on chip; writes per-instance contribution rows for the
//                   embedding tables (dL/drow without the l2 term) and per-tile partial gradients of
//                   the small dense parameters (W_k, b_k, h).
//   rows_apply      (rows_adam.hip) per embedding table: sum contribution rows per table row in batch
//                   order (+ reg * row per occurrence), TF1 sparse Adam over every row.
//   small_adam_dense the small parameters: tile partials added in tile order, TF1 dense ApplyAdam.
// Everything is deterministic (fixed summation orders, no atomics).
#include "crec_internal.h"
#include "rows_adam.h"
#include "ncf_common.h"

namespace {

constexpr int T = 32;        // instances per tile
constexpr int NT = 256;      // threads per workgroup


__device__ inline float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }

// dynamic LDS: activations A_0..A_L (A_0 = concat(pm, qm)), deltas D_1..D_L, gmf rows, per-instance scalars
__global__ __launch_bounds__(NT) void ncf_tile(NcfDev m, const int32_t *__restrict__ u, const int32_t *__restrict__ i,
                                               const float *__restrict__ y, int nb, float *__restrict__ c_pg,
                                               float *__restrict__ c_qg, float *__restrict__ c_pm,
                                               float *__restrict__ c_qm, float *__restrict__ loss_inst,
                                               float *__restrict__ partial)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, tile = blockIdx.x;
    const int r0 = tile * T;
    const int nr = (nb - r0) < T ? (nb - r0) : T;
    const bool has_g = m.kind != 1, has_m = m.kind != 0;
    const int dg = has_g ? m.dg : 0, L = has_m ? m.n_layers : 0, hm = m.L0 / 2;

    // carve LDS
    float *p = lds;
    float *A[MAX_LAYERS + 1], *D[MAX_LAYERS + 1];
    int width[MAX_LAYERS + 1];
    width[0] = has_m ? m.L0 : 0;
    for (int k = 0; k < L; ++k)
        width[k + 1] = width[k] / 2;
    for (int k = 0; k <= L; ++k) {
        A[k] = p;
        p += T * width[k];
    }
    for (int k = 1; k <= L; ++k) {
        D[k] = p;
        p += T * width[k];
    }
    float *G_p = p;
    p += T * dg;
    float *G_q = p;
    p += T * dg;
    float *zg = p, *e = p + T, *yl = p + 2 * T;
    int *su = (int *)(p + 3 * T), *si = su + T;

    for (int r = tid; r < T; r += NT) {
        const bool ok = r < nr;
        su[r] = ok ? u[r0 + r] : 0;
        si[r] = ok ? i[r0 + r] : 0;
        yl[r] = ok ? y[r0 + r] : 0.f;
        zg[r] = 0.f;
    }
    __syncthreads();
    // ---- gathers ------------------------------------------------------------------------------
    for (int k = tid; k < T * dg; k += NT) {
        const int r = k / dg, c = k % dg;
        G_p[k] = r < nr ? m.Pg[(int64_t)su[r] * dg + c] : 0.f;
        G_q[k] = r < nr ? m.Qg[(int64_t)si[r] * dg + c] : 0.f;
    }
    for (int k = tid; k < T * width[0]; k += NT) {
        const int r = k / width[0], c = k % width[0];
        float v = 0.f;
        if (r < nr)
            v = c < hm ? m.Pm[(int64_t)su[r] * hm + c] : m.Qm[(int64_t)si[r] * hm + (c - hm)];
        A[0][k] = v;
    }
    __syncthreads();
    // ---- GMF product: zg[r] = sum_c pg*qg*h_g ----------------------------------------------------
    const float *h = m.small + m.h_off; // GMF: h_gmf[dg]; MLP: h_mlp[last]; NeuMF: h_neumf[dg + last]
    if (has_g) {
        for (int r = tid >> 3; r < T; r += NT >> 3) { // 8 threads per instance
            float s = 0.f;
            for (int c = tid & 7; c < dg; c += 8)
                s = s + (G_p[r * dg + c] * G_q[r * dg + c]) * h[c];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            if ((tid & 7) == 0)
                zg[r] = s;
        }
    }
    // ---- tower forward: A[k+1] = relu(A[k] W_k + b_k) --------------------------------------------
    // W_k is staged in LDS with rows padded by one float (the backward pass reads it column-wise); each
    // thread produces 4 rows x 1 column, so a weight is read once per 4 FMAs
    float *Wl = (float *)(si + T);
    for (int k = 0; k < L; ++k) {
        const int kin = width[k], kout = width[k + 1], ws = kout + 1;
        const float *W = m.small + m.w_off[k], *b = m.small + m.b_off[k];
        __syncthreads();
        for (int o = tid; o < kin * kout; o += NT)
            Wl[(o / kout) * ws + (o % kout)] = W[o];
        __syncthreads();
        for (int unit = tid; unit < (T / 4) * kout; unit += NT) {
            const int rb = unit / kout, c = unit % kout;
            const float *a = A[k] + (rb * 4) * kin;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (int kk = 0; kk < kin; ++kk) {
                const float w = Wl[kk * ws + c];
                s0 = s0 + a[kk] * w;
                s1 = s1 + a[kin + kk] * w;
                s2 = s2 + a[2 * kin + kk] * w;
                s3 = s3 + a[3 * kin + kk] * w;
            }
            float *o4 = A[k + 1] + (rb * 4) * kout + c;
            const float bb = b[c];
            s0 += bb, s1 += bb, s2 += bb, s3 += bb;
            o4[0] = s0 > 0.f ? s0 : 0.f;
            o4[kout] = s1 > 0.f ? s1 : 0.f;
            o4[2 * kout] = s2 > 0.f ? s2 : 0.f;
            o4[3 * kout] = s3 > 0.f ? s3 : 0.f;
        }
    }
    __syncthreads();
    // ---- logit, e, loss --------------------------------------------------------------------------
    const float *h_m = h + dg;
    const int last = has_m ? width[L] : 0;
    for (int r = tid; r < T; r += NT) {
        float z = zg[r];
        for (int c = 0; c < last; ++c)
            z = z + A[L][r * last + c] * h_m[c];
        const float lab = yl[r];
        e[r] = r < nr ? sigmoidf_(z) - lab : 0.f;
        if (r < nr) {
            // sigmoid_cross_entropy_with_logits (utils/tools.py:68-69) + the l2 terms of the gathered rows
            float ls = (z > 0.f ? z : 0.f) - z * lab + log1pf(expf(-fabsf(z)));
            float ng = 0.f, nm = 0.f;
            for (int c = 0; c < dg; ++c)
                ng = ng + (G_p[r * dg + c] * G_p[r * dg + c] + G_q[r * dg + c] * G_q[r * dg + c]);
            for (int c = 0; c < width[0]; ++c)
                nm = nm + A[0][r * width[0] + c] * A[0][r * width[0] + c];
            loss_inst[r0 + r] = ls + m.reg_g * (0.5f * ng) + m.reg_m * (0.5f * nm);
        }
    }
    __syncthreads();
    float *part = partial + (int64_t)tile * m.n_small;
    // ---- GMF backward ----------------------------------------------------------------------------
    if (has_g) {
        for (int k = tid; k < T * dg; k += NT) {
            const int r = k / dg, c = k % dg;
            if (r < nr) {
                c_pg[(int64_t)(r0 + r) * dg + c] = e[r] * (G_q[k] * h[c]);
                c_qg[(int64_t)(r0 + r) * dg + c] = e[r] * (G_p[k] * h[c]);
            }
        }
        for (int c = tid; c < dg; c += NT) { // dL/dh_g
            float s = 0.f;
            for (int r = 0; r < nr; ++r)
                s = s + e[r] * (G_p[r * dg + c] * G_q[r * dg + c]);
            part[m.h_off + c] = s;
        }
    }
    if (has_m) {
        for (int c = tid; c < last; c += NT) { // dL/dh_m
            float s = 0.f;
            for (int r = 0; r < nr; ++r)
                s = s + e[r] * A[L][r * last + c];
            part[m.h_off + dg + c] = s;
        }
        // D[L] = e * h_m * relu'
        for (int o = tid; o < T * last; o += NT) {
            const int r = o / last, c = o % last;
            D[L][o] = A[L][o] > 0.f ? e[r] * h_m[c] : 0.f;
        }
        for (int k = L - 1; k >= 0; --k) {
            const int kin = width[k], kout = width[k + 1];
            const float *W = m.small + m.w_off[k];
            __syncthreads();
            // dW_k[kk][c] = sum_r A[k][r][kk] * D[k+1][r][c];  db_k[c] = sum_r D[k+1][r][c]
            for (int o = tid; o < kin * kout; o += NT) {
                const int kk = o / kout, c = o % kout;
                float s = 0.f;
                for (int r = 0; r < nr; ++r)
                    s = s + A[k][r * kin + kk] * D[k + 1][r * kout + c];
                part[m.w_off[k] + o] = s;
            }
            for (int c = tid; c < kout; c += NT) {
                float s = 0.f;
                for (int r = 0; r < nr; ++r)
                    s = s + D[k + 1][r * kout + c];
                part[m.b_off[k] + c] = s;
            }
            // delta of the layer below (k > 0: through its ReLU; k == 0: dL/dX, written over A[0]);
            // W_k back in LDS (padded rows: lanes differ in kk), 4 rows per thread
            const int ws = kout + 1;
            for (int o = tid; o < kin * kout; o += NT)
                Wl[(o / kout) * ws + (o % kout)] = W[o];
            __syncthreads();
            for (int unit = tid; unit < (T / 4) * kin; unit += NT) {
                const int rb = unit / kin, kk = unit % kin;
                const float *dr = D[k + 1] + (rb * 4) * kout, *wr = Wl + kk * ws;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                for (int c = 0; c < kout; ++c) {
                    const float w = wr[c];
                    s0 = s0 + dr[c] * w;
                    s1 = s1 + dr[kout + c] * w;
                    s2 = s2 + dr[2 * kout + c] * w;
                    s3 = s3 + dr[3 * kout + c] * w;
                }
                const float sv[4] = {s0, s1, s2, s3};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int o = (rb * 4 + j) * kin + kk;
                    if (k > 0)
                        D[k][o] = A[k][o] > 0.f ? sv[j] : 0.f;
                    else
                        A[0][o] = sv[j];
                }
            }
        }
        __syncthreads();
        for (int k = tid; k < T * width[0]; k += NT) {
            const int r = k / width[0], c = k % width[0];
            if (r < nr) {
                if (c < hm)
                    c_pm[(int64_t)(r0 + r) * hm + c] = A[0][k];
                else
                    c_qm[(int64_t)(r0 + r) * hm + (c - hm)] = A[0][k];
            }
        }
    }
}

__global__ __launch_bounds__(256) void ncf_reduce_loss(const float *__restrict__ loss_inst, int64_t n_total, int32_t B,
                                                       float *__restrict__ loss_per_batch)
{
    __shared__ double part[256];
    const int64_t t0 = (int64_t)blockIdx.x * B;
    const int nb = (int)((n_total - t0) < B ? (n_total - t0) : B);
    double s = 0.0;
    for (int k = threadIdx.x; k < nb; k += 256)
        s += (double)loss_inst[t0 + k];
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

} // namespace

extern "C" int crec_ncf_train_epoch(crec_ctx *ctx, const crec_ncf *mdl, crec_adam *opt, const int32_t *u,
                                    const int32_t *i, const float *y, int64_t N, int32_t B, float *loss_per_batch,
                                    void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, mdl && opt && N >= 0, "null argument");
    CREC_REQUIRE(ctx, N == 0 || (u && i && y && loss_per_batch), "null argument");
    CREC_REQUIRE(ctx, mdl->kind >= CREC_NCF_GMF && mdl->kind <= CREC_NCF_NEUMF, "ncf kind");
    CREC_REQUIRE(ctx, opt->adam_mode == CREC_ADAM_TF1, "the NCF step implements adam_mode tf1");
    CREC_REQUIRE(ctx, B >= 1 && B <= 16384, "batch_size %d: at most 16384", B);
    const bool has_g = mdl->kind != CREC_NCF_MLP, has_m = mdl->kind != CREC_NCF_GMF;
    NcfDev d{};
    if (int rc = ncf_describe(ctx, mdl, &d))
        return rc;
    if (N == 0)
        return CREC_OK;
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t nb = crec_ceil_div(N, B);
    const int n_tiles = (int)crec_ceil_div(B, T);
    const int64_t U = has_g ? mdl->P_gmf.rows : mdl->P_mlp.rows, I = has_g ? mdl->Q_gmf.rows : mdl->Q_mlp.rows;
    const int hm = d.L0 / 2;

    // workspace: two inverted indices (users, items) for the epoch, contribution rows, partials, losses
    const size_t ixu = crec_rows_index_bytes(nb, B, U), ixi = crec_rows_index_bytes(nb, B, I);
    const size_t cb = sizeof(float) * (size_t)B * (2 * d.dg + 2 * hm);
    const size_t pb = sizeof(float) * (size_t)n_tiles * d.n_small, lb = sizeof(float) * (size_t)nb * B;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    if (int rc = crec_reserve(ctx, &ctx->idx_ws, up(ixu) + up(ixi) + up(cb) + up(pb) + up(lb)))
        return rc;
    char *ws = (char *)ctx->idx_ws.ptr;
    crec_rows_index_view vu, vi;
    if (int rc = crec_rows_index(ctx, u, N, B, U, ws, &vu, st))
        return rc;
    if (int rc = crec_rows_index(ctx, i, N, B, I, ws + up(ixu), &vi, st))
        return rc;
    float *c_pg = (float *)(ws + up(ixu) + up(ixi));
    float *c_qg = c_pg + (size_t)B * d.dg, *c_pm = c_qg + (size_t)B * d.dg, *c_qm = c_pm + (size_t)B * hm;
    float *partial = (float *)(ws + up(ixu) + up(ixi) + up(cb));
    float *loss_inst = (float *)(ws + up(ixu) + up(ixi) + up(cb) + up(pb));

    // LDS of the tile kernel
    size_t lds_f = 0;
    {
        int w = has_m ? d.L0 : 0;
        lds_f += (size_t)T * w;
        for (int k = 0; k < d.n_layers; ++k) {
            w /= 2;
            lds_f += 2 * (size_t)T * w;
        }
        lds_f += 2 * (size_t)T * d.dg + 5 * T;
        if (has_m)
            lds_f += (size_t)d.L0 * (d.L0 / 2 + 1); // the widest layer's weights, rows padded by one float
    }
    const size_t lds = lds_f * sizeof(float);
    CREC_REQUIRE(ctx, lds <= 160 * 1024, "model too wide for the on-chip tile (%zu bytes of LDS)", lds);
    CREC_HIP(ctx, hipFuncSetAttribute((const void *)ncf_tile, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

    for (int64_t b = 0; b < nb; ++b) {
        const int64_t t0 = b * B;
        const int nbt = (int)((N - t0) < B ? (N - t0) : B);
        crec_rows_opt o{};
        o.kind = ROWS_SPARSE_TF1;
        o.lr_t = (opt->lr * sqrtf(1.0f - opt->beta2_power)) / (1.0f - opt->beta1_power);
        o.beta1 = opt->beta1;
        o.beta2 = opt->beta2;
        o.eps = opt->eps;
        crec_prof_begin(ctx, CREC_K_NCF_TILE, st);
        hipLaunchKernelGGL(ncf_tile, dim3((unsigned)crec_ceil_div(nbt, T)), dim3(NT), lds, st, d, u + t0, i + t0, y + t0, nbt,
                           c_pg, c_qg, c_pm, c_qm, loss_inst + t0, partial);
        crec_prof_end(ctx, st);
        if (has_g) {
            o.reg_occ = mdl->reg_gmf;
            crec_rows_apply(ctx, crec_rows_table{mdl->P_gmf.var, mdl->P_gmf.m, mdl->P_gmf.v, U, d.dg}, vu, b, c_pg, nullptr, o, st);
            crec_rows_apply(ctx, crec_rows_table{mdl->Q_gmf.var, mdl->Q_gmf.m, mdl->Q_gmf.v, I, d.dg}, vi, b, c_qg, nullptr, o, st);
        }
        if (has_m) {
            o.reg_occ = mdl->reg_mlp;
            crec_rows_apply(ctx, crec_rows_table{mdl->P_mlp.var, mdl->P_mlp.m, mdl->P_mlp.v, U, hm}, vu, b, c_pm, nullptr, o, st);
            crec_rows_apply(ctx, crec_rows_table{mdl->Q_mlp.var, mdl->Q_mlp.m, mdl->Q_mlp.v, I, hm}, vi, b, c_qm, nullptr, o, st);
        }
        o.kind = ROWS_DENSE;
        crec_small_adam_dense(ctx, mdl->small.var, mdl->small.m, mdl->small.v, partial, (int)crec_ceil_div(nbt, T), d.n_small,
                              d.n_small, o, st);
        opt->beta1_power = opt->beta1_power * opt->beta1;
        opt->beta2_power = opt->beta2_power * opt->beta2;
        opt->step += 1;
    }
    crec_prof_begin(ctx, CREC_K_REDUCE_LOSS, st);
    hipLaunchKernelGGL(ncf_reduce_loss, dim3((unsigned)nb), dim3(256), 0, st, (const float *)loss_inst, N, B, loss_per_batch);
    crec_prof_end(ctx, st);
    CREC_HIP(ctx, hipGetLastError());
    return crec_prof_flush(ctx, st);
}
