// Social BPR (SBPR) for gfx950.
//
// Replaces, per batch, sess.run([self.train, self.loss], {u_idx, i_idx, i_s_idx, i_neg_idx, suk})
// (model/RankingRecommender.py:101-116, train_model_sbpr) on the graph of model/ranking/SBPR.py:31-55:
//   x_ui = p_u.q_i + b_i,  x_uk = p_u.q_k + b_k (k: an item the user's friends consumed),  x_uj = p_u.q_j + b_j
//   loss = sum -log sigmoid((x_ui - x_uk) / suk) + sum -log sigmoid(x_uk - x_uj)
//          + reg * (l2(p_u) + l2(q_i) + l2(q_k) + l2(q_j) + l2(b_i) + l2(b_k) + l2(b_j))        (SBPR.py:52-54)
// with tf.train.AdamOptimizer: P, Q and bias are reached through tf.gather only, so all three take the IndexedSlices
// path -- gradients of duplicate rows summed first, then m, v decayed and var moved on EVERY row.
//
// Per batch, three launches on the machinery of rows_adam.hip (like CML / FM): sbpr_instances (a lane group per
// instance: four row gathers, three dots, the two sigmoids, contribution rows for p_u and the three item rows,
// contribution scalars for the three biases) -> crec_apply_all (P, Q, and bias as a one-column table).  The item slots
// [i | k | j] of a batch are inputs, so their index is not built per step: sbpr_slots lays out the slots of up to 64 batches
// and ONE rows_index_batches launch sorts them all (the user side's index covers the whole call).  Deterministic, no atomics.
//
// CUNE_BPR (model/ranking/CUNE_BPR.py:44-61) is the same step with other arguments of the two log-sigmoids --
//   loss = sum -log sigmoid(x_ui - x_uk) + sum -log sigmoid((x_uk - x_uj) / (s + 1.0)) + the same l2 terms
// -- where s is a trainable scalar (tf.Variable(0.0), :42; a dense variable: ApplyAdam).  sbpr_instances<LPR, true> reads
// s, leaves every instance's ds = sigmoid(-b) * b / (s + 1) (b = the second argument; RealDiv's gradient) in a column, and
// cune_scalar_step (one workgroup, launched after the tables' apply) sums the column in a fixed order and moves s.
#include "crec_internal.h"
#include "crec_store.h"
#include "rows_adam.h"

namespace {

__device__ inline float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ inline void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ inline float dot4(float4 a, float4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }

// Eigen softplus as TF1 evaluates -log_sigmoid (utils/tools.py:70-71)
__device__ inline float tf_softplus(float z)
{
    const float thr = -13.942385f; // log(FLT_EPSILON) + 2
    if (z > -thr)
        return z;
    if (z < thr)
        return expf(z);
    return log1pf(expf(z));
}

template <int LPR> __device__ inline float group_sum(float v)
{
#pragma unroll
    for (int off = LPR >> 1; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

// grid: ceil(nb / G / 4) blocks of 256; slot layout of the item side per batch: [i-slots | k-slots | j-slots], B each
template <int LPR, bool CUNE>
__global__ __launch_bounds__(256) void sbpr_instances(const float *__restrict__ P, const float *__restrict__ Q,
                                                      const float *__restrict__ bias, int d, const int32_t *__restrict__ u,
                                                      const int32_t *__restrict__ i, const int32_t *__restrict__ k,
                                                      const int32_t *__restrict__ j, const float *__restrict__ suk, int nb,
                                                      int B, float reg, float *__restrict__ c_p, float *__restrict__ c_q,
                                                      float *__restrict__ c_b, float *__restrict__ loss_inst,
                                                      const float *__restrict__ s_var,
                                                      float *__restrict__ ds_inst)
{
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const int r = (int)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * G + g);
    const bool valid = r < nb, act = 4 * l < d, on = valid && act;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int ru = valid ? u[r] : 0, ri = valid ? i[r] : 0, rk = valid ? k[r] : 0, rj = valid ? j[r] : 0;
    float sk = 1.0f, den = 1.0f; // SBPR: the first argument is divided by suk; CUNE: the second by s + 1
    if (CUNE)
        den = s_var[0] + 1.0f;
    else if (valid)
        sk = suk[r];
    const float4 p = on ? ld4(P + (int64_t)ru * d + 4 * l) : zero, qi = on ? ld4(Q + (int64_t)ri * d + 4 * l) : zero,
                 qk = on ? ld4(Q + (int64_t)rk * d + 4 * l) : zero, qj = on ? ld4(Q + (int64_t)rj * d + 4 * l) : zero;
    const float bi = valid ? bias[ri] : 0.f, bk = valid ? bias[rk] : 0.f, bj = valid ? bias[rj] : 0.f;
    const float xi = group_sum<LPR>(dot4(p, qi)) + bi, xk = group_sum<LPR>(dot4(p, qk)) + bk, xj = group_sum<LPR>(dot4(p, qj)) + bj;
    const float np = group_sum<LPR>(dot4(p, p)), ni = group_sum<LPR>(dot4(qi, qi)), nk = group_sum<LPR>(dot4(qk, qk)),
                nj = group_sum<LPR>(dot4(qj, qj));
    // SBPR.py:52: tf.divide(ui - uk, suk), uk - uj;  CUNE_BPR.py:58: ui - uk, (uk - uj) / (s + 1.0)
    const float a = CUNE ? xi - xk : (xi - xk) / sk, b = CUNE ? (xk - xj) / den : xk - xj;
    const float s1 = 1.0f / (1.0f + expf(a)), s2 = 1.0f / (1.0f + expf(b)); // sigmoid(-a), sigmoid(-b)
    const float g1 = CUNE ? s1 : s1 / sk, g2 = CUNE ? s2 / den : s2;
    const float gi = -g1, gk = g1 - g2, gj = g2; // dL/dx_ui, dL/dx_uk, dL/dx_uj
    if (valid && l == 0) {
        if (CUNE)
            ds_inst[r] = (s2 * b) / den; // dL/db = -s2;  d b / d den = -b / den (RealDiv's gradient: grad * (-x / y) / y)
        loss_inst[r] = (tf_softplus(-a) + tf_softplus(-b)) +
                       reg * (((0.5f * np + 0.5f * ni) + (0.5f * nk + 0.5f * nj)) + ((0.5f * (bi * bi) + 0.5f * (bk * bk)) + 0.5f * (bj * bj)));
        c_b[r] = gi + reg * bi; // the bias is a gathered variable too: its l2 term counts once per occurrence
        c_b[B + r] = gk + reg * bk;
        c_b[2 * B + r] = gj + reg * bj;
    }
    if (!on)
        return;
    // contribution rows WITHOUT the l2 term (rows_adam.hip adds reg * row per occurrence)
    st4_out(c_p, (int64_t)r * d + 4 * l, make_float4((gi * qi.x + gk * qk.x) + gj * qj.x, (gi * qi.y + gk * qk.y) + gj * qj.y,
                                                  (gi * qi.z + gk * qk.z) + gj * qj.z, (gi * qi.w + gk * qk.w) + gj * qj.w));
    st4_out(c_q, (int64_t)r * d + 4 * l, make_float4(gi * p.x, gi * p.y, gi * p.z, gi * p.w));
    st4_out(c_q, (int64_t)(B + r) * d + 4 * l, make_float4(gk * p.x, gk * p.y, gk * p.z, gk * p.w));
    st4_out(c_q, (int64_t)(2 * B + r) * d + 4 * l, make_float4(gj * p.x, gj * p.y, gj * p.z, gj * p.w));
}

// the item slots of a group of batches, batch after batch: [i | k | j] with stride nbt (the batch's own length: a ragged last
// batch has no holes, and a batch's slot s is the row of c_q / c_b its instance kernel writes)
__global__ __launch_bounds__(256) void sbpr_slots(const int32_t *__restrict__ i, const int32_t *__restrict__ k,
                                                  const int32_t *__restrict__ j, int64_t n, int32_t B, int32_t *__restrict__ slots)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n)
        return;
    const int64_t b = t / B, t0 = b * B;
    const int64_t nbt = (n - t0) < B ? (n - t0) : B;
    int32_t *dst = slots + 3 * t0 + (t - t0);
    dst[0] = i[t];
    dst[nbt] = k[t];
    dst[2 * nbt] = j[t];
}

// one block per batch: the batch's instance losses in a fixed order, float64
__global__ __launch_bounds__(256) void sbpr_reduce_loss(const float *__restrict__ loss_inst, int64_t n_total, int32_t B,
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

// CUNE_BPR's `s`: the batch's ds column summed in a fixed order (float64: per thread, per wave by shuffles, then the 16
// wave sums in order), then the dense update of one element.  One workgroup of 1024; a thread's <= 8 loads are in flight together.
__global__ __launch_bounds__(1024) void cune_scalar_step(const float *__restrict__ ds_inst, int nb, float *__restrict__ var,
                                                         float *__restrict__ m, float *__restrict__ v, crec_rows_opt o)
{
    __shared__ double part[16];
    double s = 0.0;
    for (int r0 = 0; r0 < nb; r0 += 8192) { // 8192 terms at a time (one round for the shipped batch: the sums it always had)
        float x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = r0 + (int)threadIdx.x + q * 1024;
            x[q] = r < nb ? ds_inst[r] : 0.f;
        }
        const double sr = (((double)x[0] + (double)x[1]) + ((double)x[2] + (double)x[3])) + (((double)x[4] + (double)x[5]) + ((double)x[6] + (double)x[7]));
        s = r0 == 0 ? sr : s + sr;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x != 0)
        return;
    s = 0.0;
    for (int w = 0; w < 16; ++w)
        s += part[w];
    const float g = (float)s;
    if (o.opt == CREC_OPT_SGD) { // GradientDescentOptimizer
        var[0] -= o.lr_t * g;
    } else if (o.opt == CREC_OPT_ADAGRAD) { // accumulator in m (0.1 at start)
        const float acc = m[0] + g * g;
        m[0] = acc;
        var[0] -= (o.lr_t * g) / sqrtf(acc);
    } else { // ApplyAdam: m += (g - m)(1 - b1); v += (g^2 - v)(1 - b2); var -= m lr_t / (sqrt(v) + eps)
        const float mn = m[0] + (g - m[0]) * (1.0f - o.beta1), vn = v[0] + (g * g - v[0]) * (1.0f - o.beta2);
        m[0] = mn;
        v[0] = vn;
        var[0] -= (mn * o.lr_t) / (sqrtf(vn) + o.eps);
    }
}

} // namespace

static int sbpr_epoch(crec_ctx *ctx, const crec_sbpr *m, const crec_param *s_par, crec_adam *opt, const int32_t *u,
                      const int32_t *i, const int32_t *k, const int32_t *j, const float *suk, int64_t N, int32_t B,
                      float *loss_per_batch, void *stream);

extern "C" int crec_sbpr_train_epoch(crec_ctx *ctx, const crec_sbpr *m, crec_adam *opt, const int32_t *u, const int32_t *i,
                                     const int32_t *k, const int32_t *j, const float *suk, int64_t N, int32_t B,
                                     float *loss_per_batch, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, N <= 0 || suk, "null argument");
    return sbpr_epoch(ctx, m, nullptr, opt, u, i, k, j, suk, N, B, loss_per_batch, stream);
}

extern "C" int crec_cune_train_epoch(crec_ctx *ctx, const crec_cune *m, crec_adam *opt, const int32_t *u, const int32_t *i,
                                     const int32_t *k, const int32_t *j, int64_t N, int32_t B, float *loss_per_batch, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, m && m->s.var && m->s.m && m->s.v && m->s.rows == 1 && m->s.cols == 1, "CUNE_BPR: s is one float with its two slots");
    const crec_sbpr tables{m->P, m->Q, m->bias, m->reg};
    return sbpr_epoch(ctx, &tables, &m->s, opt, u, i, k, j, nullptr, N, B, loss_per_batch, stream);
}

static int sbpr_epoch(crec_ctx *ctx, const crec_sbpr *m, const crec_param *s_par, crec_adam *opt, const int32_t *u,
                      const int32_t *i, const int32_t *k, const int32_t *j, const float *suk, int64_t N, int32_t B,
                      float *loss_per_batch, void *stream)
{
    const bool cune = s_par != nullptr;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, m && opt && m->P.var && m->Q.var && m->bias.var && m->P.cols == m->Q.cols && m->bias.cols == 1 &&
                          m->bias.rows >= m->Q.rows,
                 "SBPR variables: P [U,d], Q [I,d], bias [>= I, 1]");
    const int d = m->P.cols;
    CREC_REQUIRE(ctx, d % 4 == 0 && d >= 4 && d <= 256, "embed_size %d: need a multiple of 4 in [4, 256]", d);
    CREC_REQUIRE(ctx, B >= 1 && B <= 65536, "batch_size <= 65536"); // (a batch's slots are indexed 16 384 at a time)
    const int u_chunks = (int)crec_ceil_div(B, 16384);
    CREC_REQUIRE(ctx, N >= 0 && (N == 0 || (u && i && k && j && loss_per_batch)), "null argument");
    if (N == 0)
        return CREC_OK;
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t U = m->P.rows, I = m->Q.rows, nb = crec_ceil_div(N, B);
    int lpr = 1;
    while (lpr * 4 < d)
        lpr <<= 1;
    const int G = 64 / lpr;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    // item slots: the index is built ahead for many batches at once, so what counts is the walk of the update launch (a row's
    // slots chunk by chunk), not the latency of one sort: as few chunks as a workgroup's 16 384 slots allow
    const int q_chunks = (int)crec_ceil_div(3 * (int64_t)B, 16384);
    const int64_t GB = nb < 64 ? nb : 64; // batches per item-side index launch
    const size_t ixu = crec_rows_index_bytes(nb, B, U, u_chunks), ixq = crec_rows_index_bytes(GB, 3 * B, I, q_chunks);
    const size_t cpb = sizeof(float) * (size_t)B * d, cqb = sizeof(float) * (size_t)3 * B * d, cbb = sizeof(float) * (size_t)3 * B;
    const size_t sl = sizeof(int32_t) * (size_t)3 * B * GB, li = sizeof(float) * (size_t)N, dsb = sizeof(float) * (size_t)B;
    if (int rc = crec_reserve(ctx, &ctx->idx_ws, up(ixu) + up(ixq) + up(cpb) + up(cqb) + up(cbb) + up(sl) + up(li) + up(dsb)))
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
    float *c_b = (float *)p;
    p += up(cbb);
    int32_t *slots = (int32_t *)p;
    p += up(sl);
    float *loss_inst = (float *)p;
    p += up(li);
    float *ds_inst = (float *)p;
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t t0 = b * B, bg = b % GB;
        const int nbt = (int)((N - t0) < B ? (N - t0) : B);
        if (bg == 0) { // the item-side index of the next GB batches
            const int64_t ng = (N - t0) < GB * B ? (N - t0) : GB * B;
            hipLaunchKernelGGL(sbpr_slots, dim3((unsigned)crec_ceil_div(ng, 256)), dim3(256), 0, st, i + t0, k + t0, j + t0, ng, B, slots);
            if (int rc = crec_rows_index(ctx, slots, 3 * ng, 3 * B, I, ixq_ws, &vq, st, q_chunks))
                return rc;
        }
        crec_rows_opt o{};
        o.kind = ROWS_SPARSE_TF1;
        o.reg_occ = m->reg;
        if (int rc = crec_rows_optimizer(ctx, opt, &o, true))
            return rc;
        const dim3 grid((unsigned)crec_ceil_div(crec_ceil_div(nbt, G), 4));
#define CREC_SBPR_LAUNCH(L, C)                                                                                              \
    hipLaunchKernelGGL((sbpr_instances<L, C>), grid, dim3(256), 0, st, (const float *)m->P.var, (const float *)m->Q.var,    \
                       (const float *)m->bias.var, d, u + t0, i + t0, k + t0, j + t0, C ? nullptr : suk + t0, nbt, nbt,     \
                       m->reg, c_p, c_q, c_b, loss_inst + t0, C ? (const float *)s_par->var : nullptr, ds_inst)
#define CREC_SBPR(L)                                                                                                        \
    case L:                                                                                                                 \
        if (cune)                                                                                                           \
            CREC_SBPR_LAUNCH(L, true);                                                                                      \
        else                                                                                                                \
            CREC_SBPR_LAUNCH(L, false);                                                                                     \
        break;
        switch (lpr) {
            CREC_SBPR(1)
            CREC_SBPR(2)
            CREC_SBPR(4)
            CREC_SBPR(8)
            CREC_SBPR(16)
            CREC_SBPR(32)
            CREC_SBPR(64)
        }
#undef CREC_SBPR
#undef CREC_SBPR_LAUNCH
        const int2 *q_rowseg = vq.rowseg + bg * vq.chunks * vq.rows;
        const int32_t *q_order = vq.order + bg * vq.per_batch;
        crec_apply_set set{};
        set.tables[set.n_tables++] = crec_rows_job{crec_rows_table{m->P.var, m->P.m, m->P.v, U, d}, vu.chunks,
                                                   vu.rowseg + b * vu.chunks * vu.rows, vu.order + b * vu.per_batch, c_p, nullptr, nullptr, o};
        set.tables[set.n_tables++] = crec_rows_job{crec_rows_table{m->Q.var, m->Q.m, m->Q.v, I, d}, vq.chunks, q_rowseg, q_order, c_q,
                                                   nullptr, nullptr, o};
        crec_rows_opt ob = o; // the bias: its l2 term is inside the contribution scalars; g = 0 rows still decay and move (TF1) or stay (lazy)
        ob.kind = o.kind == ROWS_SPARSE_LAZY ? ROWS_SPARSE_LAZY : ROWS_DENSE;
        ob.reg_occ = 0.f;
        ob.dense_reg = 0.f;
        set.scalar = crec_scalar_job{m->bias.var, m->bias.m, m->bias.v, I, q_rowseg, q_order, c_b, nullptr, ob, vq.chunks};
        if (int rc = crec_apply_all(ctx, set, st))
            return rc;
        if (cune) // s is read by the batch's instances only: its update can follow the tables'
            hipLaunchKernelGGL(cune_scalar_step, dim3(1), dim3(1024), 0, st, (const float *)ds_inst, nbt, (float *)s_par->var,
                               (float *)s_par->m, (float *)s_par->v, ob);
        opt->beta1_power = opt->beta1_power * opt->beta1;
        opt->beta2_power = opt->beta2_power * opt->beta2;
        opt->step += 1;
    }
    hipLaunchKernelGGL(sbpr_reduce_loss, dim3((unsigned)nb), dim3(256), 0, st, (const float *)loss_inst, N, B, loss_per_batch);
    CREC_HIP(ctx, hipGetLastError());
    return crec_prof_flush(ctx, st);
}
