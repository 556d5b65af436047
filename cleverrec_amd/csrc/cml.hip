// Collaborative Metric Learning for gfx950.
//
// Replaces sess.run([train, loss], {u_idx, i_idx, neg_items}) (model/RankingRecommender.py:90-100) on the
// graph of model/ranking/CML.py:34-70; see include/crec.h (crec_cml).
//
//   d_ui = |p_u - q_i|^2, d_un[k] = |p_u - q_neg[k]|^2                                   (CML.py:42-43)
//   loss_b = max(d_ui + margin - min_k d_un, 0) * log(rank + 1),
//   rank   = mean_k[d_ui + margin - d_un[k] > 0] * item_nums / neg_ratio   (no gradient)  (CML.py:45-53)
//   + reg * sum_{a != b} cov([Q;P])_ab                                                    (CML.py:63-70)
// The covariance term reads the full tables, so P and Q get dense gradients -> TF1 dense ApplyAdam on
// every row.  Its gradient needs no d x d product: with X centred by the column means mu,
// dL/dX_ra = (2 reg / n)(s_r - Xc_ra), s_r = sum_b Xc_rb (rows_adam.hip).
//
// Per batch, three launches:
//   cml_instances   one lane group per instance: distances, arg-min negative, WARP weight, contribution rows for
//                   p_u, q_i and the arg-min negative (whose row id is only known now).  Riding in the same launch,
//                   on extra blocks: the column means mu of [Q;P] that this step's update needs (per-block column
//                   sums; the block that finishes last adds them in block order), and the previous step's loss.
//   rows_index      the item slots of this batch, in chunks (rows_adam.h)
//   crec_apply_all  P and Q: contributions + covariance gradient, dense ApplyAdam; leaves the rows' loss terms.
#include "crec_internal.h"
#include "rows_adam.h"

namespace {

__device__ inline float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ inline void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }

template <int LPR> __device__ inline float gsum(float v)
{
#pragma unroll
    for (int off = LPR >> 1; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

__device__ inline float sqdist4(float4 a, float4 b)
{
    const float x = a.x - b.x, y = a.y - b.y, z = a.z - b.z, w = a.w - b.w;
    return (x * x + y * y) + (z * z + w * w);
}

constexpr int CS_ROWS = 128; // table rows per column-sum block

struct CmlAux {
    // column means: blocks [n_inst, n_inst + n_cs) of the launch; the first nbQ of them walk Q, the others P
    const float *P, *Q;
    int64_t U, I;
    int n_inst, n_cs, nbQ;
    float *partial;    // [n_cs][d]
    unsigned *counter; // zero between launches
    float *mu;         // [d]
    float n_rows;      // U + I
    // loss of the previous step on one more block (out == nullptr: none): its instance terms and the row terms its
    // update launch left;  loss = sum_b loss_inst + (reg / n) * (sum_r rowP + sum_r rowQ)
    const float *prev_inst;
    int prev_n;
    const float *rowP, *rowQ;
    float reg_over_n;
    float *out;
};

// thread t adds elements t, t + 256, ... in that order, 8 loads in flight at a time
__device__ __forceinline__ double strided_sum(const float *__restrict__ x, int64_t n, double acc)
{
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
}

__device__ __forceinline__ void cml_loss_block(const CmlAux &a)
{
    __shared__ double part[256];
    const double s = strided_sum(a.prev_inst, a.prev_n, 0.0);
    const double c = strided_sum(a.rowQ, a.I, strided_sum(a.rowP, a.U, 0.0));
    part[threadIdx.x] = s + (double)a.reg_over_n * c;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0)
        *a.out = (float)part[0];
}

// column sums of CS_ROWS rows: LPR lanes (float4 each) along a row, 256 / LPR rows side by side; each lane adds its
// rows top to bottom, the side-by-side sums are added left to right; the block that finishes last turns the
// per-block sums into means, in block order
template <int LPR> __device__ __forceinline__ void cml_colsum_block(const CmlAux &a, int d, int blk)
{
    constexpr int RL = 256 / LPR;
    __shared__ float4 red[256];
    __shared__ int is_last;
    const int l = threadIdx.x % LPR, rl = threadIdx.x / LPR;
    const bool inQ = blk < a.nbQ, act = 4 * l < d;
    const float *X = inQ ? a.Q : a.P;
    const int64_t rows = inQ ? a.I : a.U, r0 = (int64_t)(inQ ? blk : blk - a.nbQ) * CS_ROWS;
    const int64_t r1 = (r0 + CS_ROWS) < rows ? (r0 + CS_ROWS) : rows;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act) {
        int64_t r = r0 + rl;
        for (; r + 7 * RL < r1; r += 8 * RL) {
            float4 t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                t[j] = ld4(X + (r + j * RL) * d + 4 * l);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                s = make_float4(s.x + t[j].x, s.y + t[j].y, s.z + t[j].z, s.w + t[j].w);
        }
        for (; r < r1; r += RL) {
            const float4 t = ld4(X + r * d + 4 * l);
            s = make_float4(s.x + t.x, s.y + t.y, s.z + t.z, s.w + t.w);
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0 && act) {
        for (int k = 1; k < RL; ++k) {
            const float4 t = red[k * LPR + l];
            s = make_float4(s.x + t.x, s.y + t.y, s.z + t.z, s.w + t.w);
        }
        st4(a.partial + (int64_t)blk * d + 4 * l, s);
    }
    __threadfence(); // this block's sums are visible device-wide before its ticket is
    __syncthreads();
    if (threadIdx.x == 0)
        is_last = atomicAdd(a.counter, 1u) == (unsigned)(a.n_cs - 1);
    __syncthreads();
    if (!is_last)
        return;
    __threadfence();
    for (int c = threadIdx.x; c < d; c += 256) {
        float g = 0.f;
        int p = 0;
        for (; p + 8 <= a.n_cs; p += 8) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                t[j] = a.partial[(int64_t)(p + j) * d + c];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                g = g + t[j];
        }
        for (; p < a.n_cs; ++p)
            g = g + a.partial[(int64_t)p * d + c];
        a.mu[c] = g / a.n_rows;
    }
    if (threadIdx.x == 0)
        *a.counter = 0u;
}

template <int LPR>
__global__ __launch_bounds__(256) void cml_instances(const float *__restrict__ P, const float *__restrict__ Q, int d,
                                                     const int32_t *__restrict__ u, const int32_t *__restrict__ i,
                                                     const int32_t *__restrict__ neg, int n, int K, float margin,
                                                     float rank_scale /* item_nums / neg_ratio */, float *__restrict__ c_p,
                                                     float *__restrict__ c_q, int32_t *__restrict__ item_slots,
                                                     float *__restrict__ loss_inst, CmlAux aux)
{
    constexpr int G = 64 / LPR;
    if ((int)blockIdx.x >= aux.n_inst) { // the riders: column means for this step's update, loss of the previous step
        if ((int)blockIdx.x < aux.n_inst + aux.n_cs)
            cml_colsum_block<LPR>(aux, d, (int)blockIdx.x - aux.n_inst);
        else
            cml_loss_block(aux);
        return;
    }
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const int64_t inst = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * G + g;
    const bool ok = inst < n, act = 4 * l < d;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int uu = ok ? u[inst] : 0, ii = ok ? i[inst] : 0;
    // (K <= LPR, the shipped shape: lane l of the group asks for the instance's l-th negative id here, beside the rows of u and i --
    // one coalesced load instead of a round trip of id loads in front of every batch of negative rows)
    const bool ids_up_front = K <= LPR;
    const int my_nk = (ids_up_front && ok && l < K) ? neg[inst * K + l] : 0;
    const float4 p = (ok && act) ? ld4(P + (int64_t)uu * d + 4 * l) : zero;
    const float4 qi = (ok && act) ? ld4(Q + (int64_t)ii * d + 4 * l) : zero;
    const float d_ui = gsum<LPR>(sqdist4(p, qi));
    float dmin = INFINITY;
    int kmin = 0, imp = 0;
    if (ids_up_front) {
        // ten negatives' rows in flight per group (the launch is 3 072 waves at batch 6144: registers are not what it lacks)
        constexpr int NF = 10;
        for (int k0 = 0; k0 < K; k0 += NF) { // wave-uniform trip count
            int nk[NF];
            float4 qn[NF];
#pragma unroll
            for (int e = 0; e < NF; ++e)
                nk[e] = __shfl(my_nk, g * LPR + ((k0 + e) < K ? (k0 + e) : 0), 64);
#pragma unroll
            for (int e = 0; e < NF; ++e)
                qn[e] = (ok && act && k0 + e < K) ? ld4(Q + (int64_t)nk[e] * d + 4 * l) : zero;
#pragma unroll
            for (int e = 0; e < NF; ++e) {
                if (k0 + e < K) { // wave-uniform
                    const float dn = gsum<LPR>(sqdist4(p, qn[e]));
                    if (dn < dmin) { // first minimum wins
                        dmin = dn;
                        kmin = nk[e];
                    }
                    imp += (d_ui + margin - dn) > 0.f;
                }
            }
        }
    }
    // four negatives' rows in flight per group: the loop is a chain of row fetches otherwise (K is 20 in the shipped conf)
#ifndef CML_NB
#define CML_NB 4
#endif
    constexpr int NB = CML_NB;
    for (int k0 = 0; k0 < K && !ids_up_front; k0 += NB) { // wave-uniform trip count
        int nk[NB];
        float4 qn[NB];
#pragma unroll
        for (int e = 0; e < NB; ++e)
            nk[e] = (ok && k0 + e < K) ? neg[inst * K + k0 + e] : 0;
#pragma unroll
        for (int e = 0; e < NB; ++e)
            qn[e] = (ok && act) ? ld4(Q + (int64_t)nk[e] * d + 4 * l) : zero;
#pragma unroll
        for (int e = 0; e < NB; ++e) {
            if (k0 + e < K) { // wave-uniform
                const float dn = gsum<LPR>(sqdist4(p, qn[e]));
                if (dn < dmin) { // first minimum wins
                    dmin = dn;
                    kmin = nk[e];
                }
                imp += (d_ui + margin - dn) > 0.f;
            }
        }
    }
    const float base = fmaxf(d_ui + margin - dmin, 0.f);
    const float rank = ((float)imp / (float)K) * rank_scale;
    const float wgt = logf(rank + 1.0f);
    if (!ok)
        return;
    if (l == 0) {
        loss_inst[inst] = base * wgt;
        item_slots[2 * inst] = ii;
        item_slots[2 * inst + 1] = kmin;
    }
    if (act) {
        const float4 qm = ld4(Q + (int64_t)kmin * d + 4 * l);
        const float w2 = base > 0.f ? 2.0f * wgt : 0.f;
        // d/dp = w2*((p-qi) - (p-qm)); d/dqi = -w2*(p-qi); d/dqm = +w2*(p-qm)
        const float4 a = make_float4(p.x - qi.x, p.y - qi.y, p.z - qi.z, p.w - qi.w);
        const float4 b = make_float4(p.x - qm.x, p.y - qm.y, p.z - qm.z, p.w - qm.w);
        st4(c_p + inst * d + 4 * l, make_float4(w2 * (a.x - b.x), w2 * (a.y - b.y), w2 * (a.z - b.z), w2 * (a.w - b.w)));
        st4(c_q + (2 * inst) * d + 4 * l, make_float4(-w2 * a.x, -w2 * a.y, -w2 * a.z, -w2 * a.w));
        st4(c_q + (2 * inst + 1) * d + 4 * l, make_float4(w2 * b.x, w2 * b.y, w2 * b.z, w2 * b.w));
    }
}

__global__ __launch_bounds__(256) void cml_reduce_loss(CmlAux a) { cml_loss_block(a); }

// rs-evaluation tables (CML.py:76,84; RankingRecommender.py:222-223): |clip_1(p) - q|^2 ascending ==
// (-2 clip_1(p)) . q + |q|^2 ascending, i.e. a dot product of vectors widened by one slot (padded to 4):
//   P_aug = [-2 clip_1(p), 1, 0, 0, 0],  Q_aug = [q, |q|^2, 0, 0, 0]
__global__ void cml_aug_tables(const float *__restrict__ P, const float *__restrict__ Q, int64_t U, int64_t I, int d,
                               int clip_users, float *__restrict__ Pa, float *__restrict__ Qa)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int da = d + 4;
    if (r < U) {
        float nn = 0.f;
        for (int c = 0; c < d; ++c)
            nn = nn + P[r * d + c] * P[r * d + c];
        // tf.clip_by_norm(., 1.0, axes=[1]) on the rs branch only (CML.py:76,84); the loo branch scores with the unclipped row
        const float nrm = sqrtf(nn), sc = (clip_users && nrm > 1.0f) ? 1.0f / nrm : 1.0f;
        for (int c = 0; c < d; ++c)
            Pa[r * da + c] = -2.0f * (P[r * d + c] * sc);
        Pa[r * da + d] = 1.0f;
        Pa[r * da + d + 1] = Pa[r * da + d + 2] = Pa[r * da + d + 3] = 0.f;
    } else if (r < U + I) {
        const int64_t q = r - U;
        float nn = 0.f;
        for (int c = 0; c < d; ++c) {
            const float v = Q[q * d + c];
            Qa[q * da + c] = v;
            nn = nn + v * v;
        }
        Qa[q * da + d] = nn;
        Qa[q * da + d + 1] = Qa[q * da + d + 2] = Qa[q * da + d + 3] = 0.f;
    }
}

} // namespace

extern "C" {

int crec_cml_train_epoch(crec_ctx *ctx, const crec_cml *m, crec_adam *opt, const int32_t *u, const int32_t *i,
                         const int32_t *neg, int64_t N, int32_t K, int32_t B, float *loss_per_batch, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, m && opt && m->P.var && m->Q.var && m->P.cols == m->Q.cols, "CML tables");
    const int d = m->P.cols;
    CREC_REQUIRE(ctx, d % 4 == 0 && d >= 4 && d <= 256, "embed_size %d: need a multiple of 4 in [4, 256]", d);
    // (user slots are indexed 16 384 at a time, the per-step item index 2048 at a time in at most 64 chunks)
    CREC_REQUIRE(ctx, K >= 1 && B >= 1 && B <= 65536, "neg_ratio >= 1, batch_size <= 65536");
    const int u_chunks = (int)crec_ceil_div(B, 16384);
    CREC_REQUIRE(ctx, N >= 0 && (N == 0 || (u && i && neg && loss_per_batch)), "null argument");
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
    const int nbP = (int)crec_ceil_div(U, CS_ROWS), nbQ = (int)crec_ceil_div(I, CS_ROWS);
#ifndef CML_CHUNK_SLOTS
#define CML_CHUNK_SLOTS 2048
#endif
    const int q_chunks = (int)crec_ceil_div(2 * (int64_t)B, CML_CHUNK_SLOTS); // item slots: sorted 2048 at a time
    const size_t ixu = crec_rows_index_bytes(nb, B, U, u_chunks), ixq = crec_rows_index_bytes(1, 2 * B, I, q_chunks);
    const size_t cpb = sizeof(float) * (size_t)B * d, cqb = sizeof(float) * (size_t)2 * B * d;
    const size_t sl = sizeof(int32_t) * (size_t)2 * B, li = sizeof(float) * (size_t)B;
    const size_t pp = sizeof(float) * (size_t)(nbP + nbQ) * d, mub = sizeof(float) * 256 + 256;
    const size_t rP = sizeof(float) * (size_t)U, rQ = sizeof(float) * (size_t)I;
    if (int rc = crec_reserve(ctx, &ctx->idx_ws,
                              up(ixu) + up(ixq) + up(cpb) + up(cqb) + up(sl) + 2 * up(li) + up(pp) + up(mub) + up(rP) + up(rQ)))
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
    float *loss_inst = (float *)p; // double-buffered: the next step's launch reduces this step's terms
    p += 2 * up(li);
    float *partial = (float *)p;
    p += up(pp);
    float *mu = (float *)p;
    unsigned *counter = (unsigned *)(p + sizeof(float) * 256);
    p += up(mub);
    float *rowP = (float *)p;
    p += up(rP);
    float *rowQ = (float *)p;
    const float n_rows = (float)(U + I);
    CREC_HIP(ctx, hipMemsetAsync(counter, 0, sizeof(unsigned), st));

    CmlAux aux{};
    aux.P = m->P.var, aux.Q = m->Q.var, aux.U = U, aux.I = I;
    aux.n_cs = nbP + nbQ, aux.nbQ = nbQ;
    aux.partial = partial, aux.counter = counter, aux.mu = mu, aux.n_rows = n_rows;
    aux.rowP = rowP, aux.rowQ = rowQ, aux.reg_over_n = m->reg / n_rows;
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t t0 = b * B;
        const int nbt = (int)((N - t0) < B ? (N - t0) : B);
        crec_rows_opt o{};
        o.kind = ROWS_DENSE;
        o.cov_coef = 2.0f * m->reg / n_rows;
        if (int rc = crec_rows_optimizer(ctx, opt, &o))
            return rc;
        const int waves = (int)crec_ceil_div(nbt, G);
        float *li_b = (float *)((char *)loss_inst + (b & 1) * up(li));
        aux.n_inst = (int)crec_ceil_div(waves, 4);
        const dim3 grid((unsigned)(aux.n_inst + aux.n_cs + (aux.out ? 1 : 0)));
        crec_prof_begin(ctx, CREC_K_CML_INSTANCES, st);
#define CREC_CML(L)                                                                                                          \
    case L:                                                                                                                  \
        CREC_KLAUNCH(ctx, (cml_instances<L>), grid, dim3(256), 0, st, (const float *)m->P.var, (const float *)m->Q.var, d,  \
                           u + t0, i + t0, neg + t0 * K, nbt, (int)K, m->margin, (float)I / (float)K, c_p, c_q, slots,        \
                           li_b, aux);                                                                                       \
        break;
        switch (lpr) {
            CREC_CML(1)
            CREC_CML(2)
            CREC_CML(4)
            CREC_CML(8)
            CREC_CML(16)
            CREC_CML(32)
            CREC_CML(64)
        }
#undef CREC_CML
        crec_prof_end(ctx, st);
        if (int rc = crec_rows_index(ctx, slots, 2 * (int64_t)nbt, 2 * B, I, ixq_ws, &vq, st, q_chunks))
            return rc;
        // (the item table first: its popular rows' waves are the launch's longest and must not begin in its second round of waves)
        crec_apply_set set{};
        set.tables[set.n_tables++] = crec_rows_job{crec_rows_table{m->Q.var, m->Q.m, m->Q.v, I, d}, vq.chunks, vq.rowseg, vq.order,
                                                   c_q, mu, rowQ, o};
        set.tables[set.n_tables++] = crec_rows_job{crec_rows_table{m->P.var, m->P.m, m->P.v, U, d}, vu.chunks,
                                                   vu.rowseg + b * vu.chunks * vu.rows, vu.order + b * vu.per_batch, c_p, mu, rowP, o};
        if (int rc = crec_apply_all(ctx, set, st))
            return rc;
        aux.prev_inst = li_b, aux.prev_n = nbt, aux.out = loss_per_batch + b;
        opt->beta1_power = opt->beta1_power * opt->beta1;
        opt->beta2_power = opt->beta2_power * opt->beta2;
        opt->step += 1;
    }
    hipLaunchKernelGGL(cml_reduce_loss, dim3(1), dim3(256), 0, st, aux); // the last step's loss
    CREC_HIP(ctx, hipGetLastError());
    return crec_prof_flush(ctx, st);
}

int crec_cml_eval_tables(crec_ctx *ctx, const float *P, const float *Q, int64_t U, int64_t I, int32_t d, int32_t clip_users,
                         float *P_aug, float *Q_aug, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, P && Q && P_aug && Q_aug && U >= 1 && I >= 1 && d >= 1, "bad argument");
    CREC_REQUIRE(ctx, d + 4 <= 256, "embed_size %d: the widened rows [d + 4] must fit crec_eval_topk's 256 columns", d);
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(cml_aug_tables, dim3((unsigned)crec_ceil_div(U + I, 256)), dim3(256), 0, (hipStream_t)stream, P, Q, U, I,
                       d, (int)clip_users, P_aug, Q_aug);
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}

} // extern "C"
