// Generic "sum per-instance contribution rows into table rows, then Adam" machinery shared by the
// pointwise (GMF / MLP / NeuMF), CML and FM steps.  A model's instance kernel writes, for every batch
// slot that indexes a table, the row's gradient contribution WITHOUT the l2 term; this file turns
// (slot -> row) into a per-row inverted index (TF1's unique + unsorted_segment_sum, deterministic:
// slots are summed in batch order) and applies TF1 Adam in place:
//
//   ROWS_SPARSE_TF1  IndexedSlices path (_apply_sparse_shared): m = m*b1 + g*(1-b1), v likewise, and
//                    m, v, var are swept on EVERY row of the table (g = 0 where untouched).
//   ROWS_SPARSE_LAZY the same formula on touched rows only.
//   ROWS_DENSE       ApplyAdam: m += (g-m)*(1-b1); v += (g*g-v)*(1-b2); var -= (m*lr_t)/(sqrt(v)+eps),
//                    every row; g = contributions + dense_reg*var (+ an optional dense gradient).
//
// In place is safe: the instance kernel has already consumed the old rows.
#include "crec_internal.h"
#include "rows_adam.h"

namespace {

constexpr int IDX_THREADS = 1024;

__device__ inline int lb_key(const unsigned long long *keys, int n, unsigned long long key)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (keys[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// one workgroup per batch: sort (row, slot) keys in LDS; order[k] = slot of the k-th sorted key;
// rowseg[r] = {first k of row r, number of slots on row r} for every table row
__global__ __launch_bounds__(IDX_THREADS) void rows_index_batches(const int32_t *__restrict__ idx, int64_t n_total,
                                                                  int32_t per_batch, int64_t rows,
                                                                  int32_t *__restrict__ order, int2 *__restrict__ rowseg)
{
    extern __shared__ unsigned long long keys[];
    const int batch = blockIdx.x, tid = threadIdx.x;
    const int64_t t0 = (int64_t)batch * per_batch;
    const int n = (int)((n_total - t0) < per_batch ? (n_total - t0) : per_batch);
    int n_pad = 64;
    while (n_pad < n)
        n_pad <<= 1;
    for (int k = tid; k < n_pad; k += IDX_THREADS)
        keys[k] = k < n ? (((unsigned long long)(uint32_t)idx[t0 + k] << 32) | (uint32_t)k) : ~0ull;
    __syncthreads();
    for (int k = 2; k <= n_pad; k <<= 1) {
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int t = tid; t < (n_pad >> 1); t += IDX_THREADS) {
                int lo = ((t & ~(jj - 1)) << 1) | (t & (jj - 1)), hi = lo + jj;
                unsigned long long a = keys[lo], b = keys[hi];
                if ((a > b) == ((lo & k) == 0)) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    for (int k = tid; k < n; k += IDX_THREADS)
        order[t0 + k] = (int32_t)(uint32_t)keys[k];
    int2 *rs = rowseg + (int64_t)batch * rows;
    for (int64_t r = tid; r < rows; r += IDX_THREADS) {
        const int a = lb_key(keys, n, (unsigned long long)r << 32);
        const int b = lb_key(keys, n, (unsigned long long)(r + 1) << 32);
        rs[r] = make_int2(a, b - a);
    }
}

__device__ inline float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ inline void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

// one wave per table row; group g of the wave takes slots g, g+G, ... of the row (fixed order),
// the G partial sums are added in group order
template <int LPR>
__global__ __launch_bounds__(256) void rows_apply(crec_rows_table t, const int2 *__restrict__ rowseg,
                                                  const int32_t *__restrict__ order, const float *__restrict__ contrib,
                                                  const float *__restrict__ dense_grad, float *__restrict__ row_out,
                                                  crec_rows_opt o)
{
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= t.rows)
        return;
    const int d = t.d;
    const bool act = 4 * l < d;
    const int2 rs = rowseg[r];
    if (rs.y == 0 && o.kind == ROWS_SPARSE_LAZY)
        return;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t ro = r * d + 4 * l;
    const float4 var = act ? ld4(t.var + ro) : zero;
    float4 acc = zero;
    for (int k = rs.x + g; k < rs.x + rs.y; k += G) {
        if (act) {
            const float4 c = ld4(contrib + (int64_t)order[k] * d + 4 * l);
            acc.x = acc.x + (c.x + o.reg_occ * var.x);
            acc.y = acc.y + (c.y + o.reg_occ * var.y);
            acc.z = acc.z + (c.z + o.reg_occ * var.z);
            acc.w = acc.w + (c.w + o.reg_occ * var.w);
        }
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) { // groups hold disjoint slots: add them, lower group first
        const float x = __shfl_down(acc.x, off, 64), y = __shfl_down(acc.y, off, 64);
        const float z = __shfl_down(acc.z, off, 64), w = __shfl_down(acc.w, off, 64);
        acc.x = acc.x + x;
        acc.y = acc.y + y;
        acc.z = acc.z + z;
        acc.w = acc.w + w;
    }
    // CML covariance term (CML.py:63-70): with X = [Q;P] centred by the column means mu,
    // d/dX_ra of reg * sum_{a != b} cov_ab = (2*reg/n) * (s_r - Xc_ra), s_r = sum_b Xc_rb; the loss itself is
    // (reg/n) * sum_r (s_r^2 - sum_a Xc_ra^2).  dense_grad then carries mu [d] instead of a per-row gradient.
    float4 dg = zero;
    if (o.cov_coef != 0.f) {
        float4 xc = zero;
        if (act) {
            const float4 mu = ld4(dense_grad + 4 * l);
            xc = make_float4(var.x - mu.x, var.y - mu.y, var.z - mu.z, var.w - mu.w);
        }
        float sr = (xc.x + xc.y) + (xc.z + xc.w), sq = (xc.x * xc.x + xc.y * xc.y) + (xc.z * xc.z + xc.w * xc.w);
#pragma unroll
        for (int off = LPR >> 1; off > 0; off >>= 1) {
            sr += __shfl_xor(sr, off, 64);
            sq += __shfl_xor(sq, off, 64);
        }
        dg = make_float4(o.cov_coef * (sr - xc.x), o.cov_coef * (sr - xc.y), o.cov_coef * (sr - xc.z), o.cov_coef * (sr - xc.w));
        if (row_out && lane == 0)
            row_out[r] = sr * sr - sq;
    } else if (dense_grad && act && g == 0) {
        dg = ld4(dense_grad + ro);
    }
    if (g != 0 || !act)
        return;
    float4 m = ld4(t.m + ro), v = ld4(t.v + ro), nv = var;
    float *pg = &acc.x, *pm = &m.x, *pv = &v.x, *pn = &nv.x;
    const float *pvar = &var.x;
    const float *pdg = &dg.x;
    const float omb1 = 1.0f - o.beta1, omb2 = 1.0f - o.beta2;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (o.kind == ROWS_DENSE) {
            const float gg = (pg[c] + pdg[c]) + o.dense_reg * pvar[c];
            pm[c] = pm[c] + (gg - pm[c]) * omb1;
            pv[c] = pv[c] + (gg * gg - pv[c]) * omb2;
            pn[c] = pvar[c] - (pm[c] * o.lr_t) / (sqrtf(pv[c]) + o.eps);
        } else {
            pm[c] = pm[c] * o.beta1 + pg[c] * omb1;
            pv[c] = pv[c] * o.beta2 + (pg[c] * pg[c]) * omb2;
            pn[c] = pvar[c] - (o.lr_t * pm[c]) / (sqrtf(pv[c]) + o.eps);
        }
    }
    st4(t.var + ro, nv);
    st4(t.m + ro, m);
    st4(t.v + ro, v);
}

// dense ApplyAdam of a small parameter vector (h, W_k, b_k, w0) from partial sums:
//   g[e] = sum_p partial[p*stride + e] (p ascending), n elements; one thread per element
__global__ void small_adam_dense(float *__restrict__ var, float *__restrict__ m, float *__restrict__ v,
                                 const float *__restrict__ partial, int n_partials, int64_t stride, int n,
                                 crec_rows_opt o)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n)
        return;
    float g = 0.f;
    for (int p = 0; p < n_partials; ++p)
        g = g + partial[(int64_t)p * stride + e];
    const float omb1 = 1.0f - o.beta1, omb2 = 1.0f - o.beta2;
    float mm = m[e], vv = v[e];
    mm = mm + (g - mm) * omb1;
    vv = vv + (g * g - vv) * omb2;
    m[e] = mm;
    v[e] = vv;
    var[e] = var[e] - (mm * o.lr_t) / (sqrtf(vv) + o.eps);
}

} // namespace

size_t crec_rows_index_bytes(int64_t n_batches, int32_t per_batch, int64_t rows)
{
    return sizeof(int32_t) * (size_t)n_batches * per_batch + sizeof(int2) * (size_t)n_batches * rows + 512;
}

int crec_rows_index(crec_ctx *ctx, const int32_t *idx, int64_t n_total, int32_t per_batch, int64_t rows, void *ws,
                    crec_rows_index_view *out, hipStream_t st)
{
    CREC_REQUIRE(ctx, per_batch >= 1 && per_batch <= 16384, "rows index: at most 16384 slots per batch, got %d", per_batch);
    const int64_t nb = crec_ceil_div(n_total, per_batch);
    out->order = (int32_t *)ws;
    out->rowseg = (int2 *)((char *)ws + ((sizeof(int32_t) * (size_t)nb * per_batch + 255) / 256) * 256);
    out->per_batch = per_batch;
    out->rows = rows;
    int n_pad = 64;
    while (n_pad < per_batch)
        n_pad <<= 1;
    const size_t lds = sizeof(unsigned long long) * n_pad;
    CREC_HIP(ctx, hipFuncSetAttribute((const void *)rows_index_batches, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    crec_prof_begin(ctx, CREC_K_ROWS_INDEX, st);
    hipLaunchKernelGGL(rows_index_batches, dim3((unsigned)nb), dim3(IDX_THREADS), lds, st, idx, n_total, per_batch, rows,
                       out->order, out->rowseg);
    crec_prof_end(ctx, st);
    return CREC_OK;
}

int crec_rows_apply(crec_ctx *ctx, const crec_rows_table &t, const crec_rows_index_view &ix, int64_t batch,
                    const float *contrib, const float *dense_grad, const crec_rows_opt &o, hipStream_t st, float *row_out)
{
    int lpr = 1;
    while (lpr * 4 < t.d)
        lpr <<= 1;
    const int2 *rs = ix.rowseg + batch * ix.rows;
    const int32_t *ord = ix.order + batch * ix.per_batch;
    const dim3 grid((unsigned)crec_ceil_div(t.rows, 4)), block(256);
    crec_prof_begin(ctx, CREC_K_ROWS_APPLY, st);
#define CREC_RA(L)                                                                                   \
    case L:                                                                                          \
        hipLaunchKernelGGL((rows_apply<L>), grid, block, 0, st, t, rs, ord, contrib, dense_grad, row_out, o); \
        break;
    switch (lpr) {
        CREC_RA(1)
        CREC_RA(2)
        CREC_RA(4)
        CREC_RA(8)
        CREC_RA(16)
        CREC_RA(32)
        CREC_RA(64)
    }
#undef CREC_RA
    crec_prof_end(ctx, st);
    return CREC_OK;
}

int crec_small_adam_dense(crec_ctx *ctx, float *var, float *m, float *v, const float *partial, int n_partials,
                          int64_t stride, int n, const crec_rows_opt &o, hipStream_t st)
{
    crec_prof_begin(ctx, CREC_K_SMALL_ADAM, st);
    hipLaunchKernelGGL(small_adam_dense, dim3((unsigned)crec_ceil_div(n, 256)), dim3(256), 0, st, var, m, v, partial,
                       n_partials, stride, n, o);
    crec_prof_end(ctx, st);
    return CREC_OK;
}
