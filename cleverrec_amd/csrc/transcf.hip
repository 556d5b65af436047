// Translational Collaborative Filtering (TransCF) for gfx950: training step and full-catalogue distances.
//
// Replaces, per batch, sess.run([self.train, self.loss], {u_idx, i_idx, j_idx}) (model/RankingRecommender.py:39-47) on the
// graph of model/ranking/TransCF.py:39-77 and, for evaluation, sess.run(self.pre_scores) (TransCF.py:85-91):
//   NU = A_ui Q   (a user's neighbourhood: the mean of its training items' rows; utils/tools.py:100-113, TransCF.py:42)
//   NI = A_iu P   (an item's neighbourhood: the mean of its training users' rows; TransCF.py:43)
//   r_ui = NU[u] * NI[i]          dist_ui = |p_u + r_ui - q_i|^2                                   (TransCF.py:49-54)
//   loss = sum max(x + margin, 0) + reg1 * (|p_u - NU[u]|^2 + |q_i - NI[i]|^2) + reg2 * (x + margin)^2,  x = dist_ui - dist_uj
// with tf.train.AdamOptimizer.  P and Q receive a DENSE gradient each -- through the two sparse-dense products every row of
// Q reaches the loss via the users that hold it, and TF adds the lookups' IndexedSlices to that dense tensor -- so both
// tables take the dense ApplyAdam kernel (m += (g - m)(1 - b1), ...) over every row.  _unit_clipping (TransCF.py:79-83)
// only rebinds graph tensors after the train op exists: its one observable effect is that the rs branch of _predict scores
// with clip_1(p_u) (TransCF.py:88-91), the loo branch with ui_dist as built (TransCF.py:86-87).
//
// Per batch (lane groups of d/4 lanes, float4 per lane):
//   rows_index        the batch's 2B item slots [i | j] sorted by row (rows_adam.hip); the users' once per epoch
//   tcf_nbr_rows x 2  NU rows of the batch's users, NI rows of the batch's items (rows no triple touches are skipped)
//   tcf_triples       a lane group per triple: six row loads, both distances, hinge and regularisers, and the four kinds
//                     of gradient rows per slot: direct (P, Q) and through the neighbourhoods (NU, NI)
//   tcf_row_sums x 2  per table row, its slots' rows added in slot order (fixed order) -> dense [rows, d], zero elsewhere
//   tcf_grad_adam x 2 g[row] = direct[row] + sum over the row's neighbours of w_nb * (other side's neighbourhood gradient)
//                     -- the adjoint sparse-dense product as a gather, no atomics -- then dense Adam in place
// Deterministic: fixed summation orders.
#include "crec_internal.h"
#include "rows_adam.h"

namespace {

__device__ inline float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ inline void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ inline float dot4(float4 a, float4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
__device__ inline float4 f4(float x, float y, float z, float w) { return make_float4(x, y, z, w); }

template <int LPR> __device__ inline float group_sum(float v)
{
#pragma unroll
    for (int off = LPR >> 1; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

// how many of a batch's slots name row r (index of rows_adam.hip, `chunks` sorted runs)
__device__ inline int touched(const int2 *__restrict__ rowseg, int64_t rows, int chunks, int64_t r)
{
    int n = 0;
    for (int ch = 0; ch < chunks; ++ch)
        n += rowseg[(int64_t)ch * rows + r].y;
    return n;
}

// SparseTensor values of get_sp_mat (utils/tools.py:106,110): float32(1.0 / n), n = the row's neighbour count
__device__ inline float nbr_weight(int n) { return n > 0 ? (float)(1.0 / (double)n) : 0.f; }

// Neighbour sums.  A row with up to HEAVY neighbours is one lane group's job (neighbours in CSR order, eight rows in flight);
// a longer one -- a popular item has thousands of users -- would be that group's chain of hundreds of dependent round trips
// and the launch as long as it, so such rows are skipped by the light launch and taken by a second one with a WORKGROUP per
// row: its 256 / LPR lane groups take every (256 / LPR)-th neighbour, the partial sums meet in LDS and are added in group
// order.  Either way one fixed order per row.
constexpr int HEAVY = 64;

// sum over k = k0 + gi, k0 + gi + ng, ... < k1 of w_k * table[idx[k]].  ADJ = false: w_k = w_row (the forward products; each
// term scaled first, as the CPU kernel of tf.sparse_tensor_dense_matmul does).  ADJ = true (the adjoint products): neighbour
// nb weighs aw[nb] = 1 / deg(nb) if a slot of the batch names it, else 0 -- its gradient row is zero and is not read
// (tcf_row_sums leaves aw beside the sums: one 4-byte load per neighbour decides).
template <bool ADJ>
__device__ __forceinline__ float4 nbr_sum(const float *__restrict__ table, const int32_t *__restrict__ idx, int64_t k0, int64_t k1, int gi,
                                          int ng, int d, int l, float w_row, const float *__restrict__ aw)
{
    constexpr int DEEP = 8;
    float4 acc = f4(0.f, 0.f, 0.f, 0.f);
    for (int64_t k = k0 + gi; k < k1; k += (int64_t)DEEP * ng) {
        float4 c[DEEP];
        float w[DEEP];
#pragma unroll
        for (int e = 0; e < DEEP; ++e) {
            c[e] = f4(0.f, 0.f, 0.f, 0.f);
            w[e] = 0.f;
            const int64_t kk = k + (int64_t)e * ng;
            if (kk < k1) {
                const int nb = idx[kk];
                w[e] = ADJ ? aw[nb] : w_row;
                if (!ADJ || w[e] != 0.f)
                    c[e] = ld4(table + (int64_t)nb * d + 4 * l);
            }
        }
#pragma unroll
        for (int e = 0; e < DEEP; ++e) {
            acc.x = acc.x + w[e] * c[e].x;
            acc.y = acc.y + w[e] * c[e].y;
            acc.z = acc.z + w[e] * c[e].z;
            acc.w = acc.w + w[e] * c[e].w;
        }
    }
    return acc;
}

// the workgroup-per-row form: every thread's partial -> the row's sum, valid in lane group 0 (threads 0 .. LPR-1)
template <int LPR> __device__ __forceinline__ float4 heavy_reduce(float4 part, float4 *__restrict__ red)
{
    constexpr int NG = 256 / LPR;
    red[threadIdx.x] = part;
    __syncthreads();
    float4 acc = f4(0.f, 0.f, 0.f, 0.f);
    if ((int)threadIdx.x < LPR)
        for (int g = 0; g < NG; ++g) {
            const float4 x = red[g * LPR + threadIdx.x];
            acc = f4(acc.x + x.x, acc.y + x.y, acc.z + x.z, acc.w + x.w);
        }
    return acc;
}

// out[r] = sum_k w_r * table[idx[k]] over row r's neighbours; rows no slot of the batch names are skipped.
// WG = false: a lane group per row, rows with more than HEAVY neighbours left out; WG = true: a workgroup per such row.
template <int LPR, bool WG>
__global__ __launch_bounds__(256) void tcf_nbr_rows(float *__restrict__ out, const float *__restrict__ table, const int64_t *__restrict__ ptr,
                                                    const int32_t *__restrict__ idx, int64_t rows, int d, const int2 *__restrict__ rowseg,
                                                    int chunks)
{
    __shared__ float4 red[WG ? 256 : 1];
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const int64_t r = WG ? (int64_t)blockIdx.x : ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * G + g;
    if (r >= rows)
        return;
    const int64_t k0 = ptr[r], k1 = ptr[r + 1];
    if ((k1 - k0 > HEAVY) != WG) // (workgroup-uniform when WG)
        return;
    if (rowseg && touched(rowseg, rows, chunks, r) == 0)
        return;
    const float w = nbr_weight((int)(k1 - k0));
    const bool act = 4 * l < d;
    float4 acc = f4(0.f, 0.f, 0.f, 0.f);
    if (act)
        acc = nbr_sum<false>(table, idx, k0, k1, WG ? (int)threadIdx.x / LPR : 0, WG ? 256 / LPR : 1, d, l, w, nullptr);
    if (WG) {
        acc = heavy_reduce<LPR>(acc, red);
        if ((int)threadIdx.x >= LPR)
            return;
    }
    if (act)
        st4(out + r * d + 4 * l, acc);
}

// a lane group per triple.  Slot layout of the item side: [i-slots | j-slots], nb each.
template <int LPR>
__global__ __launch_bounds__(256) void tcf_triples(const float *__restrict__ P, const float *__restrict__ Q, const float *__restrict__ NU,
                                                   const float *__restrict__ NI, int d, const int32_t *__restrict__ u,
                                                   const int32_t *__restrict__ i, const int32_t *__restrict__ j, int nb, float reg1,
                                                   float reg2, float margin, float *__restrict__ c_p, float *__restrict__ c_nu,
                                                   float *__restrict__ c_q, float *__restrict__ c_ni, float *__restrict__ loss_inst)
{
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const int r = (int)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * G + g);
    const bool valid = r < nb, on = valid && 4 * l < d;
    const float4 zero = f4(0.f, 0.f, 0.f, 0.f);
    const int ru = valid ? u[r] : 0, ri = valid ? i[r] : 0, rj = valid ? j[r] : 0;
    const int64_t ou = (int64_t)ru * d + 4 * l, oi = (int64_t)ri * d + 4 * l, oj = (int64_t)rj * d + 4 * l;
    const float4 p = on ? ld4(P + ou) : zero, qi = on ? ld4(Q + oi) : zero, qj = on ? ld4(Q + oj) : zero;
    const float4 nu = on ? ld4(NU + ou) : zero, ni = on ? ld4(NI + oi) : zero, nj = on ? ld4(NI + oj) : zero;
    // f = (p_u + NU * NI) - q  (TransCF.py:49-54)
    const float4 fi = f4((p.x + nu.x * ni.x) - qi.x, (p.y + nu.y * ni.y) - qi.y, (p.z + nu.z * ni.z) - qi.z, (p.w + nu.w * ni.w) - qi.w);
    const float4 fj = f4((p.x + nu.x * nj.x) - qj.x, (p.y + nu.y * nj.y) - qj.y, (p.z + nu.z * nj.z) - qj.z, (p.w + nu.w * nj.w) - qj.w);
    const float4 du = f4(p.x - nu.x, p.y - nu.y, p.z - nu.z, p.w - nu.w), di = f4(qi.x - ni.x, qi.y - ni.y, qi.z - ni.z, qi.w - ni.w);
    const float dist_i = group_sum<LPR>(dot4(fi, fi)), dist_j = group_sum<LPR>(dot4(fj, fj));
    const float rn = group_sum<LPR>(dot4(du, du)) + group_sum<LPR>(dot4(di, di)); // neighbourhood regulariser (TransCF.py:69)
    const float hm = (dist_i - dist_j) + margin;                                   // hinge argument; also ui_dist + margin - uj_dist (:71)
    if (valid && l == 0)
        loss_inst[r] = ((hm > 0.f ? hm : 0.f) + reg1 * rn) + reg2 * (hm * hm);
    if (!on)
        return;
    // dL/ddist_i = [hm >= 0] + 2 reg2 hm = -dL/ddist_j (tf.maximum passes the gradient where y + margin >= 0)
    const float gx = (hm >= 0.f ? 1.0f : 0.f) + 2.0f * reg2 * hm, a = 2.0f * gx, b = 2.0f * reg1;
    const float4 gi = f4(a * fi.x, a * fi.y, a * fi.z, a * fi.w), gj = f4(-a * fj.x, -a * fj.y, -a * fj.z, -a * fj.w); // dL/df_i, dL/df_j
    const int64_t s_u = (int64_t)r * d + 4 * l, s_i = s_u, s_j = (int64_t)(nb + r) * d + 4 * l;
    st4(c_p + s_u, f4((gi.x + gj.x) + b * du.x, (gi.y + gj.y) + b * du.y, (gi.z + gj.z) + b * du.z, (gi.w + gj.w) + b * du.w));
    st4(c_nu + s_u, f4((gi.x * ni.x + gj.x * nj.x) - b * du.x, (gi.y * ni.y + gj.y * nj.y) - b * du.y, (gi.z * ni.z + gj.z * nj.z) - b * du.z,
                       (gi.w * ni.w + gj.w * nj.w) - b * du.w));
    st4(c_q + s_i, f4(b * di.x - gi.x, b * di.y - gi.y, b * di.z - gi.z, b * di.w - gi.w));
    st4(c_q + s_j, f4(-gj.x, -gj.y, -gj.z, -gj.w));
    st4(c_ni + s_i, f4(gi.x * nu.x - b * di.x, gi.y * nu.y - b * di.y, gi.z * nu.z - b * di.z, gi.w * nu.w - b * di.w));
    st4(c_ni + s_j, f4(gj.x * nu.x, gj.y * nu.y, gj.z * nu.z, gj.w * nu.w));
}

// dst_a[r] = sum of ca over row r's slots, dst_b[r] = sum of cb, slot order; rows without slots are zero.
// aw[r] = the row's weight in its neighbourhood matrix, 1 / deg(r), if it has slots, else 0 (see nbr_sum)
template <int LPR>
__global__ __launch_bounds__(256) void tcf_row_sums(float *__restrict__ dst_a, float *__restrict__ dst_b, float *__restrict__ aw,
                                                    const float *__restrict__ ca, const float *__restrict__ cb,
                                                    const int2 *__restrict__ rowseg, const int32_t *__restrict__ order,
                                                    const int64_t *__restrict__ ptr, int64_t rows, int d, int chunks)
{
    constexpr int G = 64 / LPR, DEEP = 4;
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const int64_t r = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * G + g;
    if (r >= rows || 4 * l >= d)
        return;
    float4 a = f4(0.f, 0.f, 0.f, 0.f), b = a;
    int n_slots = 0;
    for (int ch = 0; ch < chunks; ++ch) {
        const int2 rs = rowseg[(int64_t)ch * rows + r];
        n_slots += rs.y;
        for (int k = rs.x; k < rs.x + rs.y; k += DEEP) {
            float4 x[DEEP], y[DEEP];
#pragma unroll
            for (int e = 0; e < DEEP; ++e) {
                const bool in = k + e < rs.x + rs.y;
                const int64_t o = in ? (int64_t)order[k + e] * d + 4 * l : 0;
                x[e] = in ? ld4(ca + o) : f4(0.f, 0.f, 0.f, 0.f);
                y[e] = in ? ld4(cb + o) : f4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int e = 0; e < DEEP; ++e)
                if (k + e < rs.x + rs.y) {
                    a = f4(a.x + x[e].x, a.y + x[e].y, a.z + x[e].z, a.w + x[e].w);
                    b = f4(b.x + y[e].x, b.y + y[e].y, b.z + y[e].z, b.w + y[e].w);
                }
        }
    }
    st4(dst_a + r * d + 4 * l, a);
    st4(dst_b + r * d + 4 * l, b);
    if (l == 0)
        aw[r] = n_slots ? nbr_weight((int)(ptr[r + 1] - ptr[r])) : 0.f;
}

// g[r] = direct[r] + sum_k w(idx[k]) * nbr_grad[idx[k]] over row r's neighbours (the adjoint of the OTHER side's product:
// neighbour nb contributes with ITS weight aw[nb]), then tf.train.AdamOptimizer's dense kernel on the
// row.  WG as in tcf_nbr_rows.
template <int LPR, bool WG>
__global__ __launch_bounds__(256) void tcf_grad_adam(float *__restrict__ var, float *__restrict__ m, float *__restrict__ v,
                                                     const float *__restrict__ direct, const float *__restrict__ nbr_grad,
                                                     const int64_t *__restrict__ ptr, const int32_t *__restrict__ idx,
                                                     const float *__restrict__ aw, int64_t rows, int d, float lr_t, float beta1,
                                                     float beta2, float eps, int opt)
{
    __shared__ float4 red[WG ? 256 : 1];
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const int64_t r = WG ? (int64_t)blockIdx.x : ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * G + g;
    if (r >= rows)
        return;
    const int64_t k0 = ptr[r], k1 = ptr[r + 1];
    if ((k1 - k0 > HEAVY) != WG)
        return;
    const bool act = 4 * l < d;
    float4 acc = f4(0.f, 0.f, 0.f, 0.f);
    if (act)
        acc = nbr_sum<true>(nbr_grad, idx, k0, k1, WG ? (int)threadIdx.x / LPR : 0, WG ? 256 / LPR : 1, d, l, 0.f, aw);
    if (WG) {
        acc = heavy_reduce<LPR>(acc, red);
        if ((int)threadIdx.x >= LPR)
            return;
    }
    if (!act)
        return;
    const int64_t ro = r * d + 4 * l;
    const float4 w0 = ld4(var + ro);
    float4 mm = ld4(m + ro), vv = ld4(v + ro);
    const float4 dd = ld4(direct + ro);
    const float4 gg = f4(acc.x + dd.x, acc.y + dd.y, acc.z + dd.z, acc.w + dd.w);
    const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
    // get_optimizer (utils/tools.py:79-87) on a DENSE gradient: Adam's ApplyAdam; GradientDescentOptimizer var -= lr g; AdagradOptimizer
    // accum += g^2 (in the m slot, 0.1 at start), var -= lr g / sqrt(accum) -- the arithmetic of rows_adam.hip's opt_elem
    const auto upd = [&](float g_, float var_, float &m_, float &v_) {
        if (opt == CREC_OPT_SGD)
            return var_ - lr_t * g_;
        if (opt == CREC_OPT_ADAGRAD) {
            m_ = m_ + g_ * g_;
#ifdef CREC_EXACT_MATH
            return var_ - (lr_t * g_) / sqrtf(m_);
#else
            return var_ - (lr_t * g_) * __builtin_amdgcn_rsqf(m_);
#endif
        }
        m_ = m_ + (g_ - m_) * omb1;
        v_ = v_ + (g_ * g_ - v_) * omb2;
        return var_ - (m_ * lr_t) / (sqrtf(v_) + eps);
    };
    float4 nv;
    nv.x = upd(gg.x, w0.x, mm.x, vv.x);
    nv.y = upd(gg.y, w0.y, mm.y, vv.y);
    nv.z = upd(gg.z, w0.z, mm.z, vv.z);
    nv.w = upd(gg.w, w0.w, mm.w, vv.w);
    st4(var + ro, nv);
    st4(m + ro, mm);
    st4(v + ro, vv);
}

// scores[b][item] = |a + NU[u] * NI[item] - q_item|^2, a = p_u or clip_1(p_u) (TransCF.py:85-91).  A workgroup owns TU users
// (their a and NU rows in LDS) and walks the items, a lane group per item: an item's two rows are read once for TU users.
constexpr int TU = 8;
template <int LPR>
__global__ __launch_bounds__(256) void tcf_scores(const float *__restrict__ P, const float *__restrict__ Q, const float *__restrict__ NU,
                                                  const float *__restrict__ NI, int d, const int32_t *__restrict__ users, int Bt, int64_t I,
                                                  int clip, float *__restrict__ scores)
{
    __shared__ float4 ua[TU * 64], ub[TU * 64]; // [TU][LPR] (LPR <= 64)
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const int b0 = blockIdx.y * TU;
    // stage the users: wave w takes users w, w + 4 (lane group 0 of the wave does the work; LPR lanes hold a row)
    for (int t = threadIdx.x >> 6; t < TU; t += (int)(blockDim.x >> 6)) {
        const bool on = b0 + t < Bt && 4 * l < d && g == 0;
        const int64_t ro = on ? (int64_t)users[b0 + t] * d + 4 * l : 0;
        float4 a = on ? ld4(P + ro) : f4(0.f, 0.f, 0.f, 0.f);
        const float4 nb = on ? ld4(NU + ro) : f4(0.f, 0.f, 0.f, 0.f);
        if (clip) { // tf.clip_by_norm(t, 1.0, axes=1): t * 1.0 / max(|t|, 1.0)
            const float n2 = group_sum<LPR>(dot4(a, a)), nrm = sqrtf(n2), den = nrm > 1.0f ? nrm : 1.0f;
            a = f4((a.x * 1.0f) / den, (a.y * 1.0f) / den, (a.z * 1.0f) / den, (a.w * 1.0f) / den);
        }
        if (g == 0) {
            ua[t * LPR + l] = a;
            ub[t * LPR + l] = nb;
        }
    }
    __syncthreads();
    const bool act = 4 * l < d;
    const int64_t stride = (int64_t)gridDim.x * (blockDim.x >> 6) * G;
    for (int64_t it = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * G + g; it < I; it += stride) {
        const float4 c = act ? ld4(NI + it * d + 4 * l) : f4(0.f, 0.f, 0.f, 0.f), q = act ? ld4(Q + it * d + 4 * l) : f4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < TU; ++t) {
            const float4 a = ua[t * LPR + l], nb = ub[t * LPR + l];
            const float4 f = f4((a.x + nb.x * c.x) - q.x, (a.y + nb.y * c.y) - q.y, (a.z + nb.z * c.z) - q.z, (a.w + nb.w * c.w) - q.w);
            const float s = group_sum<LPR>(act ? dot4(f, f) : 0.f);
            if (l == 0 && b0 + t < Bt)
                scores[(int64_t)(b0 + t) * I + it] = s;
        }
    }
}

__global__ __launch_bounds__(256) void tcf_reduce_loss(const float *__restrict__ loss_inst, int64_t n_total, int32_t B,
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

__global__ void tcf_cat_slots(const int32_t *__restrict__ i, const int32_t *__restrict__ j, int nb, int32_t *__restrict__ slots)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nb) {
        slots[t] = i[t];
        slots[nb + t] = j[t];
    }
}

int tcf_check(crec_ctx *ctx, const crec_transcf *m, int *d_out, int *lpr_out)
{
    CREC_REQUIRE(ctx, m && m->P.var && m->Q.var && m->P.cols == m->Q.cols, "TransCF variables: P [U,d], Q [I,d]");
    CREC_REQUIRE(ctx, m->ui_ptr && m->ui_items && m->iu_ptr && m->iu_users, "TransCF: the two neighbourhood CSRs (get_sp_mat)");
    const int d = m->P.cols;
    CREC_REQUIRE(ctx, d % 4 == 0 && d >= 4 && d <= 256, "embed_size %d: need a multiple of 4 in [4, 256]", d);
    int lpr = 1;
    while (lpr * 4 < d)
        lpr <<= 1;
    *d_out = d;
    *lpr_out = lpr;
    return CREC_OK;
}

#define TCF_LPR_SWITCH(lpr, CALL) \
    switch (lpr) {                \
    case 1: CALL(1); break;       \
    case 2: CALL(2); break;       \
    case 4: CALL(4); break;       \
    case 8: CALL(8); break;       \
    case 16: CALL(16); break;     \
    case 32: CALL(32); break;     \
    case 64: CALL(64); break;     \
    }

// NU / NI rows (all of them when rowseg == nullptr) into nu / ni
int tcf_neighbourhoods(crec_ctx *ctx, const crec_transcf *m, int d, int lpr, float *nu, float *ni, const int2 *u_rowseg, int u_chunks,
                       const int2 *i_rowseg, int i_chunks, hipStream_t st)
{
    const int64_t U = m->P.rows, I = m->Q.rows;
    const int G = 64 / lpr;
    const dim3 gu((unsigned)crec_ceil_div(crec_ceil_div(U, G), 4)), gi((unsigned)crec_ceil_div(crec_ceil_div(I, G), 4));
#define TCF_NBR(L)                                                                                                                         \
    hipLaunchKernelGGL((tcf_nbr_rows<L, false>), gu, dim3(256), 0, st, nu, (const float *)m->Q.var, m->ui_ptr, m->ui_items, U, d, u_rowseg, \
                       u_chunks);                                                                                                          \
    hipLaunchKernelGGL((tcf_nbr_rows<L, true>), dim3((unsigned)U), dim3(256), 0, st, nu, (const float *)m->Q.var, m->ui_ptr, m->ui_items,   \
                       U, d, u_rowseg, u_chunks);                                                                                          \
    hipLaunchKernelGGL((tcf_nbr_rows<L, false>), gi, dim3(256), 0, st, ni, (const float *)m->P.var, m->iu_ptr, m->iu_users, I, d, i_rowseg, \
                       i_chunks);                                                                                                          \
    hipLaunchKernelGGL((tcf_nbr_rows<L, true>), dim3((unsigned)I), dim3(256), 0, st, ni, (const float *)m->P.var, m->iu_ptr, m->iu_users,   \
                       I, d, i_rowseg, i_chunks)
    TCF_LPR_SWITCH(lpr, TCF_NBR)
#undef TCF_NBR
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}

} // namespace

extern "C" int crec_transcf_train_epoch(crec_ctx *ctx, const crec_transcf *m, crec_adam *opt, const int32_t *u, const int32_t *i,
                                        const int32_t *j, int64_t N, int32_t B, float *loss_per_batch, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, m && opt, "null argument");
    int d = 0, lpr = 0;
    if (int rc = tcf_check(ctx, m, &d, &lpr))
        return rc;
    CREC_REQUIRE(ctx, m->P.m && m->P.v && m->Q.m && m->Q.v, "Adam slots");
    CREC_REQUIRE(ctx, opt->adam_mode == CREC_ADAM_TF1 || opt->adam_mode == CREC_OPT_SGD || opt->adam_mode == CREC_OPT_ADAGRAD,
                 "the TransCF step implements adam_mode tf1 (Adam), CREC_OPT_SGD and CREC_OPT_ADAGRAD");
    const int opt_kind = opt->adam_mode == CREC_ADAM_TF1 ? 0 : opt->adam_mode;
    CREC_REQUIRE(ctx, B >= 1 && B <= 65536, "batch_size <= 65536"); // (user slots indexed 16 384 at a time, item slots 2048 at a time in <= 64 chunks)
    const int u_chunks = (int)crec_ceil_div(B, 16384);
    CREC_REQUIRE(ctx, N >= 0 && (N == 0 || (u && i && j && loss_per_batch)), "null argument");
    if (N == 0)
        return CREC_OK;
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t U = m->P.rows, I = m->Q.rows, nb = crec_ceil_div(N, B);
    const int G = 64 / lpr;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const int q_chunks = (int)crec_ceil_div(2 * (int64_t)B, 2048); // item slots: sorted 2048 at a time
    const size_t ixu = crec_rows_index_bytes(nb, B, U, u_chunks), ixq = crec_rows_index_bytes(1, 2 * B, I, q_chunks);
    const size_t tu = sizeof(float) * (size_t)U * d, ti = sizeof(float) * (size_t)I * d;
    const size_t cu = sizeof(float) * (size_t)B * d, cq = sizeof(float) * (size_t)2 * B * d;
    const size_t sl = sizeof(int32_t) * (size_t)2 * B, li = sizeof(float) * (size_t)N;
    const size_t awu = sizeof(float) * (size_t)U, awi = sizeof(float) * (size_t)I;
    if (int rc = crec_reserve(ctx, &ctx->idx_ws, up(ixu) + up(ixq) + 3 * up(tu) + 3 * up(ti) + 2 * up(cu) + 2 * up(cq) + up(sl) + up(li) + up(awu) + up(awi)))
        return rc;
    char *p = (char *)ctx->idx_ws.ptr;
    crec_rows_index_view vu, vq;
    if (int rc = crec_rows_index(ctx, u, N, B, U, p, &vu, st, u_chunks))
        return rc;
    p += up(ixu);
    char *ixq_ws = p;
    p += up(ixq);
    const auto take = [&](size_t bytes) {
        float *r = (float *)p;
        p += up(bytes);
        return r;
    };
    float *NU = take(tu), *GPd = take(tu), *GNU = take(tu), *NI = take(ti), *GQd = take(ti), *GNI = take(ti);
    float *c_p = take(cu), *c_nu = take(cu), *c_q = take(cq), *c_ni = take(cq);
    int32_t *slots = (int32_t *)take(sl);
    float *loss_inst = take(li);
    float *aw_u = take(awu), *aw_i = take(awi);
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t t0 = b * B;
        const int nbt = (int)((N - t0) < B ? (N - t0) : B);
        const float lr_t = opt_kind ? opt->lr : (opt->lr * sqrtf(1.0f - opt->beta2_power)) / (1.0f - opt->beta1_power);
        hipLaunchKernelGGL(tcf_cat_slots, dim3((unsigned)crec_ceil_div(nbt, 256)), dim3(256), 0, st, i + t0, j + t0, nbt, slots);
        if (int rc = crec_rows_index(ctx, slots, 2 * (int64_t)nbt, 2 * B, I, ixq_ws, &vq, st, q_chunks))
            return rc;
        const int2 *urs = vu.rowseg + b * vu.chunks * vu.rows;
        const int32_t *uord = vu.order + b * vu.per_batch;
        if (int rc = tcf_neighbourhoods(ctx, m, d, lpr, NU, NI, urs, vu.chunks, vq.rowseg, vq.chunks, st))
            return rc;
        const dim3 gt((unsigned)crec_ceil_div(crec_ceil_div(nbt, G), 4));
        const dim3 gu((unsigned)crec_ceil_div(crec_ceil_div(U, G), 4)), gi((unsigned)crec_ceil_div(crec_ceil_div(I, G), 4));
#define TCF_GA(L, WG, GU, GI)                                                                                                           \
    hipLaunchKernelGGL((tcf_grad_adam<L, WG>), GU, dim3(256), 0, st, m->P.var, m->P.m, m->P.v, (const float *)GPd, (const float *)GNI,    \
                       m->ui_ptr, m->ui_items, (const float *)aw_i, U, d, lr_t, opt->beta1, opt->beta2, opt->eps, opt_kind);             \
    hipLaunchKernelGGL((tcf_grad_adam<L, WG>), GI, dim3(256), 0, st, m->Q.var, m->Q.m, m->Q.v, (const float *)GQd, (const float *)GNU,    \
                       m->iu_ptr, m->iu_users, (const float *)aw_u, I, d, lr_t, opt->beta1, opt->beta2, opt->eps, opt_kind)
#define TCF_STEP(L)                                                                                                                      \
    hipLaunchKernelGGL((tcf_triples<L>), gt, dim3(256), 0, st, (const float *)m->P.var, (const float *)m->Q.var, (const float *)NU,        \
                       (const float *)NI, d, u + t0, i + t0, j + t0, nbt, m->reg1, m->reg2, m->margin, c_p, c_nu, c_q, c_ni,              \
                       loss_inst + t0);                                                                                                  \
    hipLaunchKernelGGL((tcf_row_sums<L>), gu, dim3(256), 0, st, GPd, GNU, aw_u, (const float *)c_p, (const float *)c_nu, urs, uord,        \
                       m->ui_ptr, U, d, vu.chunks);                                                                                      \
    hipLaunchKernelGGL((tcf_row_sums<L>), gi, dim3(256), 0, st, GQd, GNI, aw_i, (const float *)c_q, (const float *)c_ni,                   \
                       (const int2 *)vq.rowseg, (const int32_t *)vq.order, m->iu_ptr, I, d, vq.chunks);                                  \
    TCF_GA(L, false, gu, gi);                                                                                                            \
    TCF_GA(L, true, dim3((unsigned)U), dim3((unsigned)I))
        TCF_LPR_SWITCH(lpr, TCF_STEP)
#undef TCF_STEP
#undef TCF_GA
        opt->beta1_power = opt->beta1_power * opt->beta1;
        opt->beta2_power = opt->beta2_power * opt->beta2;
        opt->step += 1;
    }
    hipLaunchKernelGGL(tcf_reduce_loss, dim3((unsigned)nb), dim3(256), 0, st, (const float *)loss_inst, N, B, loss_per_batch);
    CREC_HIP(ctx, hipGetLastError());
    return crec_prof_flush(ctx, st);
}

extern "C" int crec_transcf_scores(crec_ctx *ctx, const crec_transcf *m, const int32_t *users, int32_t Bt, int32_t clip_users,
                                   float *nbr_ws, float *scores, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    int d = 0, lpr = 0;
    if (int rc = tcf_check(ctx, m, &d, &lpr))
        return rc;
    CREC_REQUIRE(ctx, Bt >= 0 && (Bt == 0 || (users && scores && nbr_ws)), "users / scores / nbr_ws");
    if (Bt == 0)
        return CREC_OK;
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t U = m->P.rows, I = m->Q.rows;
    float *NU = nbr_ws, *NI = nbr_ws + U * d; // caller's workspace: (U + I) * d floats
    if (int rc = tcf_neighbourhoods(ctx, m, d, lpr, NU, NI, nullptr, 1, nullptr, 1, st))
        return rc;
    const int G = 64 / lpr;
    int64_t gx = crec_ceil_div(crec_ceil_div(I, G), 4);
    gx = gx > 64 ? 64 : gx;
    const dim3 grid((unsigned)gx, (unsigned)crec_ceil_div(Bt, TU));
    CREC_REQUIRE(ctx, grid.y <= 65535, "at most %d users per call", 65535 * TU);
#define TCF_SC(L)                                                                                                                   \
    hipLaunchKernelGGL((tcf_scores<L>), grid, dim3(256), 0, st, (const float *)m->P.var, (const float *)m->Q.var, (const float *)NU, \
                       (const float *)NI, d, users, Bt, I, clip_users, scores)
    TCF_LPR_SWITCH(lpr, TCF_SC)
#undef TCF_SC
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}
