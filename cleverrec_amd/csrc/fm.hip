// Factorization Machine (rating path) for gfx950.
//
// Replaces sess.run([train, loss, y_pre]) / sess.run(y_pre) (model/RatingRecommender.py:50,73) on the graph
// of model/rating/FM.py:20-53 and the Python RMSE/MAE loop (utils/metrics.py:22-29); see include/crec.h.
//
//   y_pre = w0 + sum_f wi[x_f] + 0.5 * sum_k [ (sum_f x_f v_fk)^2 - sum_f x_f^2 v_fk^2 ]   (FM.py:43-49;
//           the linear term ignores x_value, FM.py:35,49)
//   loss  = sum_b (y - y_pre)^2 + reg * 0.5 * (|wi|^2 + |vif|^2) over the FULL tables (FM.py:52)
// so wi and vif receive dense gradients and take TF1's dense ApplyAdam on every row each step.
//
// Per batch, two launches: fm_instances (one lane group per instance: gathers, y_pre, residual, contribution rows
// for vif and scalars for wi, per-wave partial of dL/dw0; one extra block reduces the previous step's loss) ->
// crec_apply_all (rows_adam.hip: vif rows ROWS_DENSE + reg*table, wi as a one-column table, w0).
// Deterministic, no atomics.
#include "crec_internal.h"
#include "rows_adam.h"

namespace {

constexpr int FM_MAX_F = 8;   // features per row held in registers (the shipped data: 2)
constexpr int FM_WIDE_F = 64; // rows up to this many features: the wide form walks them twice instead of holding them

// loss_b = sum_b (y - y_pre)^2 + reg * 0.5 * (sum rownorm_w + sum rownorm_v), float64, fixed order
struct FmLossJob { // the loss of one finished step: its squared errors and the row norms its update launch left
    const float *sq_err;
    int nb;
    const float *nw, *nv;
    int64_t rows;
    float reg;
    float *out; // nullptr: nothing to reduce
};

__device__ __forceinline__ void fm_reduce_loss_block(const float *__restrict__ sq_err, int nb, const float *__restrict__ nw,
                                                     const float *__restrict__ nv, int64_t rows, float reg,
                                                     float *__restrict__ out)
{
    __shared__ double part[256];
    double s = 0.0, l2 = 0.0;
    // thread t adds elements t, t + 256, ... in that order; 8 loads in flight at a time
    const auto strided_sum = [](const float *__restrict__ x, int64_t n, double acc) {
        int64_t k = threadIdx.x;
        for (; k + 7 * 256 < n; k += 8 * 256) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                t[j] = x[k + j * 256];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                acc += (double)t[j];
        }
        for (; k < n; k += 256)
            acc += (double)x[k];
        return acc;
    };
    s = strided_sum(sq_err, nb, s);
    l2 = strided_sum(nw, rows, l2);
    l2 = strided_sum(nv, rows, l2);
    part[threadIdx.x] = s + (double)reg * 0.5 * l2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        *out = (float)part[0];
}

__global__ __launch_bounds__(256) void fm_reduce_loss(FmLossJob j)
{
    fm_reduce_loss_block(j.sq_err, j.nb, j.nw, j.nv, j.rows, j.reg, j.out);
}

// group = LPR lanes (float4 per lane) per instance; G = 64/LPR instances per wave
template <int LPR, bool TRAIN>
__global__ __launch_bounds__(256) void fm_instances(const float *__restrict__ w0, const float *__restrict__ wi,
                                                    const float *__restrict__ vif, int d, const int32_t *__restrict__ x_idx,
                                                    const float *__restrict__ x_val, const float *__restrict__ y, int n,
                                                    int F, float *__restrict__ y_pre, float *__restrict__ c_v,
                                                    float *__restrict__ c_w, float *__restrict__ sq_err,
                                                    float *__restrict__ w0_partial, FmLossJob prev)
{
    constexpr int G = 64 / LPR;
    if (TRAIN && prev.out && blockIdx.x == gridDim.x - 1) { // the extra last block reduces the previous step's loss
        fm_reduce_loss_block(prev.sq_err, prev.nb, prev.nw, prev.nv, prev.rows, prev.reg, prev.out);
        return;
    }
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t inst = wave * G + g;
    const bool ok = inst < n, act = 4 * l < d;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v[FM_MAX_F];
    float xv[FM_MAX_F];
    int xi[FM_MAX_F];
    float4 S = zero, SS = zero; // sum_f x v, sum_f x^2 v^2  (this lane's 4 components)
    float lin = 0.f;
#pragma unroll
    for (int f = 0; f < FM_MAX_F; ++f) {
        v[f] = zero;
        xv[f] = 0.f;
        xi[f] = 0;
        if (f < F && ok) {
            xi[f] = x_idx[inst * F + f];
            xv[f] = x_val[inst * F + f];
            if (act)
                v[f] = *reinterpret_cast<const float4 *>(vif + (int64_t)xi[f] * d + 4 * l);
            lin = lin + wi[xi[f]];
            const float x2 = xv[f] * xv[f];
            S.x = S.x + xv[f] * v[f].x;
            S.y = S.y + xv[f] * v[f].y;
            S.z = S.z + xv[f] * v[f].z;
            S.w = S.w + xv[f] * v[f].w;
            SS.x = SS.x + x2 * (v[f].x * v[f].x);
            SS.y = SS.y + x2 * (v[f].y * v[f].y);
            SS.z = SS.z + x2 * (v[f].z * v[f].z);
            SS.w = SS.w + x2 * (v[f].w * v[f].w);
        }
    }
    float second = ((S.x * S.x - SS.x) + (S.y * S.y - SS.y)) + ((S.z * S.z - SS.z) + (S.w * S.w - SS.w));
#pragma unroll
    for (int off = LPR >> 1; off > 0; off >>= 1)
        second += __shfl_xor(second, off, 64);
    const float yp = (w0[0] + lin) + 0.5f * second;
    if (ok && l == 0)
        y_pre[inst] = yp;
    if (!TRAIN)
        return;
    const float res = ok ? yp - y[inst] : 0.f; // dL/dy_pre = 2 * res
    const float gy = 2.0f * res;
    if (ok) {
        if (l == 0)
            sq_err[inst] = res * res;
#pragma unroll
        for (int f = 0; f < FM_MAX_F; ++f) {
            if (f < F) {
                if (act) { // d y / d v_fk = x_f * S_k - x_f^2 * v_fk
                    const float x2 = xv[f] * xv[f];
                    float4 c;
                    c.x = gy * (xv[f] * S.x - x2 * v[f].x);
                    c.y = gy * (xv[f] * S.y - x2 * v[f].y);
                    c.z = gy * (xv[f] * S.z - x2 * v[f].z);
                    c.w = gy * (xv[f] * S.w - x2 * v[f].w);
                    *reinterpret_cast<float4 *>(c_v + (inst * F + f) * d + 4 * l) = c;
                }
                if (l == 0)
                    c_w[inst * F + f] = gy;
            }
        }
    }
    // dL/dw0 partial of this wave: sum over its G instances in group order
    float s = l == 0 ? gy : 0.f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_xor(s, off, 64);
    if (lane == 0)
        w0_partial[wave] = s;
}

// the same for rows of more than FM_MAX_F features: nothing is held across the row -- the forward pass sums S, SS and the linear
// term feature by feature in the same order, the backward pass reads the rows again (they are in cache)
template <int LPR, bool TRAIN>
__global__ __launch_bounds__(256) void fm_instances_wide(const float *__restrict__ w0, const float *__restrict__ wi,
                                                         const float *__restrict__ vif, int d, const int32_t *__restrict__ x_idx,
                                                         const float *__restrict__ x_val, const float *__restrict__ y, int n,
                                                         int F, float *__restrict__ y_pre, float *__restrict__ c_v,
                                                         float *__restrict__ c_w, float *__restrict__ sq_err,
                                                         float *__restrict__ w0_partial, FmLossJob prev)
{
    constexpr int G = 64 / LPR;
    if (TRAIN && prev.out && blockIdx.x == gridDim.x - 1) {
        fm_reduce_loss_block(prev.sq_err, prev.nb, prev.nw, prev.nv, prev.rows, prev.reg, prev.out);
        return;
    }
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t inst = wave * G + g;
    const bool ok = inst < n, act = 4 * l < d;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 S = zero, SS = zero;
    float lin = 0.f;
    if (ok)
        for (int f = 0; f < F; ++f) {
            const int xi = x_idx[inst * F + f];
            const float xv = x_val[inst * F + f], x2 = xv * xv;
            const float4 v = act ? *reinterpret_cast<const float4 *>(vif + (int64_t)xi * d + 4 * l) : zero;
            lin = lin + wi[xi];
            S.x = S.x + xv * v.x, S.y = S.y + xv * v.y, S.z = S.z + xv * v.z, S.w = S.w + xv * v.w;
            SS.x = SS.x + x2 * (v.x * v.x), SS.y = SS.y + x2 * (v.y * v.y), SS.z = SS.z + x2 * (v.z * v.z), SS.w = SS.w + x2 * (v.w * v.w);
        }
    float second = ((S.x * S.x - SS.x) + (S.y * S.y - SS.y)) + ((S.z * S.z - SS.z) + (S.w * S.w - SS.w));
#pragma unroll
    for (int off = LPR >> 1; off > 0; off >>= 1)
        second += __shfl_xor(second, off, 64);
    const float yp = (w0[0] + lin) + 0.5f * second;
    if (ok && l == 0)
        y_pre[inst] = yp;
    if (!TRAIN)
        return;
    const float res = ok ? yp - y[inst] : 0.f;
    const float gy = 2.0f * res;
    if (ok) {
        if (l == 0)
            sq_err[inst] = res * res;
        for (int f = 0; f < F; ++f) {
            if (act) {
                const int xi = x_idx[inst * F + f];
                const float xv = x_val[inst * F + f], x2 = xv * xv;
                const float4 v = *reinterpret_cast<const float4 *>(vif + (int64_t)xi * d + 4 * l);
                float4 c;
                c.x = gy * (xv * S.x - x2 * v.x), c.y = gy * (xv * S.y - x2 * v.y), c.z = gy * (xv * S.z - x2 * v.z), c.w = gy * (xv * S.w - x2 * v.w);
                *reinterpret_cast<float4 *>(c_v + (inst * F + f) * d + 4 * l) = c;
            }
            if (l == 0)
                c_w[inst * F + f] = gy;
        }
    }
    float s = l == 0 ? gy : 0.f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_xor(s, off, 64);
    if (lane == 0)
        w0_partial[wave] = s;
}

// utils/metrics.py:22-29 in float64: out = {sqrt(mean (y - y_pre)^2), mean |y - y_pre|}
__global__ __launch_bounds__(1024) void rmse_mae_kernel(const float *__restrict__ y, const float *__restrict__ yp, int64_t n,
                                                        double *__restrict__ out)
{
    __shared__ double sq[1024], ab[1024];
    double a = 0.0, b = 0.0;
    for (int64_t k = threadIdx.x; k < n; k += 1024) {
        const double r = (double)y[k] - (double)yp[k];
        a += r * r;
        b += fabs(r);
    }
    sq[threadIdx.x] = a;
    ab[threadIdx.x] = b;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            sq[threadIdx.x] += sq[threadIdx.x + off];
            ab[threadIdx.x] += ab[threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = sqrt(sq[0] / (double)n);
        out[1] = ab[0] / (double)n;
    }
}

template <bool TRAIN>
void launch_instances(crec_ctx *ctx, int lpr, dim3 grid, hipStream_t st, const float *w0, const float *wi, const float *vif, int d,
                      const int32_t *x_idx, const float *x_val, const float *y, int n, int F, float *y_pre, float *c_v,
                      float *c_w, float *sq_err, float *w0_partial, FmLossJob prev = FmLossJob{})
{
#define CREC_FM(L)                                                                                                      \
    case L:                                                                                                             \
        if (F <= FM_MAX_F)                                                                                              \
            CREC_KLAUNCH(ctx, (fm_instances<L, TRAIN>), grid, dim3(256), 0, st, w0, wi, vif, d, x_idx, x_val, y, n, F, y_pre, \
                               c_v, c_w, sq_err, w0_partial, prev);                                                     \
        else                                                                                                            \
            CREC_KLAUNCH(ctx, (fm_instances_wide<L, TRAIN>), grid, dim3(256), 0, st, w0, wi, vif, d, x_idx, x_val, y, n, F, y_pre, \
                               c_v, c_w, sq_err, w0_partial, prev);                                                     \
        break;
    switch (lpr) {
        CREC_FM(1)
        CREC_FM(2)
        CREC_FM(4)
        CREC_FM(8)
        CREC_FM(16)
        CREC_FM(32)
        CREC_FM(64)
    }
#undef CREC_FM
}

int check_fm(crec_ctx *ctx, const crec_fm *m, int F)
{
    CREC_REQUIRE(ctx, m && m->w0.var && m->wi.var && m->vif.var, "null FM parameter");
    CREC_REQUIRE(ctx, m->vif.cols % 4 == 0 && m->vif.cols >= 4 && m->vif.cols <= 256,
                 "embed_size %d: need a multiple of 4 in [4, 256]", m->vif.cols);
    CREC_REQUIRE(ctx, m->wi.rows * m->wi.cols == m->vif.rows, "wi and vif must have the same number of features");
    CREC_REQUIRE(ctx, F >= 1 && F <= FM_WIDE_F, "%d features per row: this build takes at most %d", F, FM_WIDE_F);
    return CREC_OK;
}

} // namespace

extern "C" {

int crec_fm_predict(crec_ctx *ctx, const crec_fm *m, const int32_t *x_idx, const float *x_val, int64_t N, int32_t F,
                    float *y_pre, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    if (int rc = check_fm(ctx, m, F))
        return rc;
    CREC_REQUIRE(ctx, N >= 0 && (N == 0 || (x_idx && x_val && y_pre)), "null argument");
    if (N == 0)
        return CREC_OK;
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int d = m->vif.cols;
    int lpr = 1;
    while (lpr * 4 < d)
        lpr <<= 1;
    const int G = 64 / lpr;
    const int64_t waves = crec_ceil_div(N, G);
    launch_instances<false>(ctx, lpr, dim3((unsigned)crec_ceil_div(waves, 4)), st, m->w0.var, m->wi.var, m->vif.var, d, x_idx, x_val,
                            nullptr, (int)N, F, y_pre, nullptr, nullptr, nullptr, nullptr);
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}

int crec_fm_train_epoch(crec_ctx *ctx, const crec_fm *m, crec_adam *opt, const int32_t *x_idx, const float *x_val,
                        const float *y, int64_t N, int32_t F, int32_t B, float *y_pre, float *loss_per_batch,
                        void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    if (int rc = check_fm(ctx, m, F))
        return rc;
    CREC_REQUIRE(ctx, opt && N >= 0 && (N == 0 || (x_idx && x_val && y && y_pre && loss_per_batch)), "null argument");
    // (a batch's B * F slots are indexed 16 384 at a time; more of them in several chunks the update launch walks in order)
    CREC_REQUIRE(ctx, B >= 1 && (int64_t)B * F <= 1048576, "batch_size * features = %lld: at most 1048576", (long long)B * F);
    const int ix_chunks = (int)crec_ceil_div((int64_t)B * F, 16384);
    if (N == 0)
        return CREC_OK;
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int d = m->vif.cols;
    const int64_t rows = m->vif.rows, nb = crec_ceil_div(N, B);
    int lpr = 1;
    while (lpr * 4 < d)
        lpr <<= 1;
    const int G = 64 / lpr;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t ixb = crec_rows_index_bytes(nb, B * F, rows, ix_chunks);
    const size_t cvb = sizeof(float) * (size_t)B * F * d, cwb = sizeof(float) * (size_t)B * F, seb = sizeof(float) * (size_t)B;
    const int max_waves = (int)crec_ceil_div(B, G);
    const size_t w0b = sizeof(float) * (size_t)max_waves, nrb = sizeof(float) * (size_t)rows;
    if (int rc = crec_reserve(ctx, &ctx->idx_ws, up(ixb) + up(cvb) + up(cwb) + 2 * up(seb) + up(w0b) + 2 * up(nrb)))
        return rc;
    char *ws = (char *)ctx->idx_ws.ptr;
    crec_rows_index_view vx;
    if (int rc = crec_rows_index(ctx, x_idx, N * F, B * F, rows, ws, &vx, st, ix_chunks))
        return rc;
    float *c_v = (float *)(ws + up(ixb));
    float *c_w = (float *)((char *)c_v + up(cvb));
    float *sq_err = (float *)((char *)c_w + up(cwb));
    float *w0_partial = (float *)((char *)sq_err + 2 * up(seb)); // sq_err is double-buffered (see below)
    float *nw = (float *)((char *)w0_partial + up(w0b));
    float *nv = (float *)((char *)nw + up(nrb));

    // Two launches per step: fm_instances (+ one extra block that reduces the PREVIOUS step's loss: its squared
    // errors sit in the other half of the sq_err double buffer, and nw / nv still hold the row norms its update
    // launch wrote) and crec_apply_all (vif rows, wi as a one-column table, w0).
    FmLossJob prev{};
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t t0 = b * B;
        const int nbt = (int)((N - t0) < B ? (N - t0) : B);
        crec_rows_opt o{};
        o.kind = ROWS_DENSE;
        o.reg_occ = 0.f;
        o.dense_reg = m->reg;
        if (int rc = crec_rows_optimizer(ctx, opt, &o))
            return rc;
        const int waves = (int)crec_ceil_div(nbt, G);
        float *se = (float *)((char *)sq_err + (b & 1) * up(seb));
        crec_prof_begin(ctx, CREC_K_FM_INSTANCES, st);
        launch_instances<true>(ctx, lpr, dim3((unsigned)crec_ceil_div(waves, 4) + (prev.out ? 1u : 0u)), st, m->w0.var, m->wi.var,
                               m->vif.var, d, x_idx + t0 * F, x_val + t0 * F, y + t0, nbt, F, y_pre + t0, c_v, c_w, se,
                               w0_partial, prev);
        crec_prof_end(ctx, st);
        crec_apply_set set{};
        crec_rows_opt ov = o;
        ov.row_norm = 1;
        set.tables[set.n_tables++] = crec_rows_job{crec_rows_table{m->vif.var, m->vif.m, m->vif.v, rows, d}, vx.chunks,
                                                   vx.rowseg + b * vx.chunks * vx.rows, vx.order + b * vx.per_batch, c_v, nullptr, nv, ov};
        set.scalar = crec_scalar_job{m->wi.var, m->wi.m, m->wi.v, rows, vx.rowseg + b * vx.chunks * vx.rows, vx.order + b * vx.per_batch,
                                     c_w, nw, o, vx.chunks};
        // waves beyond this batch's count were not written: pass the exact count
        set.small = crec_small_job{m->w0.var, m->w0.m, m->w0.v, w0_partial, waves, 1, 1, o};
        if (int rc = crec_apply_all(ctx, set, st))
            return rc;
        prev = FmLossJob{se, nbt, nw, nv, rows, m->reg, loss_per_batch + b};
        opt->beta1_power = opt->beta1_power * opt->beta1;
        opt->beta2_power = opt->beta2_power * opt->beta2;
        opt->step += 1;
    }
    hipLaunchKernelGGL(fm_reduce_loss, dim3(1), dim3(256), 0, st, prev); // the last step's loss
    CREC_HIP(ctx, hipGetLastError());
    return crec_prof_flush(ctx, st);
}

int crec_rmse_mae(crec_ctx *ctx, const float *y, const float *y_pre, int64_t n, double *out2, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, y && y_pre && out2 && n >= 1, "bad argument");
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(rmse_mae_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, y, y_pre, n, out2);
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}

} // extern "C"
