// BPR training step for gfx950: gather -> pairwise loss -> per-row gradient sums -> Adam,
// without atomics and without ever materialising gradient rows in HBM.
//
// Replaces, per batch, sess.run([train, loss]) on the graph of model/ranking/BPR.py:31-44
// with tf.train.AdamOptimizer semantics (utils/tools.py:83-84); see include/crec.h.
//
// Per chunk of batches, one launch:
//   bpr_index_batches  one workgroup per (batch, side): bitonic sort in LDS of (row, position)
//                      keys -> unique rows, segment starts, occurrence records {u, i, j, pos}.
//                      This is the "unique + unsorted_segment_sum" bookkeeping of TF1's sparse
//                      Adam, done once for the whole chunk, in batch position order, i-slices
//                      before j-slices, so every row sum has one fixed order (deterministic).
//                      tf1 mode also writes row -> {segment start, length} for every table row.
// Per batch:
//   tf1 mode   ONE launch, bpr_step_fused: every row of P and Q moves every step (TF1 decays
//              m, v and applies var -= lr_t*m/(sqrt(v)+eps) on all rows), so the step reads
//              the tables from one buffer set and writes the other (ping-pong): no gather can
//              see a row updated by the same step, and user-row and item-row waves run side by
//              side.  An item wave recomputes s_b = sigmoid(-x_b) of its occurrences from
//              p_u, q_i, q_j with the very instruction sequence the user wave uses, so both
//              sides use bit-identical s_b.
//   lazy mode  two launches in place: bpr_user_side (stores the contribution row c_b = s_b*p_u
//              of each triple, plain coalesced stores, then Adam on the user's row) and
//              bpr_item_side (sums -/+ c_b over the row's occurrences, Adam on the row).
//              Rows are read only by their own wave inside a launch, so in place is safe.
// Per chunk: bpr_reduce_loss sums the per-triple loss terms of each batch in a fixed order.
//
// A wave owns one table row.  It splits into G = 64/LPR groups of LPR lanes, each lane holding
// a float4 of a row (LPR*4 >= d): G occurrences are processed at once and every row access is
// one coalesced 16-byte-per-lane load.  The wave fetches up to 64 occurrence records with one
// coalesced load and hands them to its groups by cross-lane reads; row gathers of up to UNR
// sub-iterations are issued before the first is consumed.  Dots reduce with DPP adds.
//
// Compiled with -ffp-contract=off: each fp32 multiply/add is rounded separately, like the
// element-wise Eigen kernels TF1 runs (and like the CPU oracle), so the Adam arithmetic is
// bit-reproducible.
#include "crec_internal.h"

namespace {

constexpr int PREP_THREADS = 1024;
constexpr int MAX_B = 8192;
constexpr int UNR = 4; // sub-iterations whose gathers are in flight together

// ------------------------------------------------------------------------------------
// index workspace layout for one chunk of nbc batches
// ------------------------------------------------------------------------------------
struct IndexWs {
    int32_t *n_u, *n_q;        // [nbc] unique rows per batch
    int32_t *u_rows, *u_seg;   // [nbc][B], [nbc][B+1]
    int4 *u_occ;               // [nbc][B]   {u, i, j, position in batch}
    int32_t *q_rows, *q_seg;   // [nbc][2B], [nbc][2B+1]
    int4 *q_occ;               // [nbc][2B]  {u, i, j, position | side<<31}  (side 1: row is the triple's j)
    int2 *u_rowseg, *q_rowseg; // tf1 mode: [nbc][U], [nbc][I]  {segment start, length} of every table row
    int64_t B, U, I;
    size_t bytes;
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

IndexWs layout_index(void *base, int64_t nbc, int64_t B, int64_t U, int64_t I, bool dense)
{
    IndexWs w{};
    w.B = B;
    w.U = dense ? U : 0;
    w.I = dense ? I : 0;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? (char *)base + off : nullptr;
        off = align_up(off + bytes, 256);
        return (void *)p;
    };
    w.n_u = (int32_t *)take(sizeof(int32_t) * nbc);
    w.n_q = (int32_t *)take(sizeof(int32_t) * nbc);
    w.u_rows = (int32_t *)take(sizeof(int32_t) * nbc * B);
    w.u_seg = (int32_t *)take(sizeof(int32_t) * nbc * (B + 1));
    w.u_occ = (int4 *)take(sizeof(int4) * nbc * B);
    w.q_rows = (int32_t *)take(sizeof(int32_t) * nbc * 2 * B);
    w.q_seg = (int32_t *)take(sizeof(int32_t) * nbc * (2 * B + 1));
    w.q_occ = (int4 *)take(sizeof(int4) * nbc * 2 * B);
    w.u_rowseg = (int2 *)take(sizeof(int2) * nbc * w.U);
    w.q_rowseg = (int2 *)take(sizeof(int2) * nbc * w.I);
    w.bytes = off;
    return w;
}

// ------------------------------------------------------------------------------------
// block-wide exclusive scan of one int per thread (PREP_THREADS threads)
// ------------------------------------------------------------------------------------
__device__ inline int wave_incl_scan(int v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(v, off, 64);
        if (lane >= off)
            v += t;
    }
    return v;
}

__device__ inline int block_excl_scan(int v, int *wave_sums /* [PREP_THREADS/64 + 1] LDS */, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = wave_incl_scan(v);
    if (lane == 63)
        wave_sums[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        int s = lane < PREP_THREADS / 64 ? wave_sums[lane] : 0;
        int si = wave_incl_scan(s);
        if (lane < PREP_THREADS / 64)
            wave_sums[lane] = si - s;
        if (lane == PREP_THREADS / 64 - 1)
            wave_sums[PREP_THREADS / 64] = si;
    }
    __syncthreads();
    *total = wave_sums[PREP_THREADS / 64];
    return wave_sums[wave] + incl - v;
}

__device__ inline int lower_bound_key(const unsigned long long *keys, int n, unsigned long long key)
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

// ------------------------------------------------------------------------------------
// bpr_index_batches: grid = 2*nbc workgroups; block b>>1 = batch, b&1 = side (0 user, 1 item)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(PREP_THREADS) void bpr_index_batches(const int32_t *__restrict__ u,
                                                                  const int32_t *__restrict__ i,
                                                                  const int32_t *__restrict__ j, int64_t n_total,
                                                                  int32_t B, IndexWs w)
{
    extern __shared__ unsigned long long keys[]; // [n_pad], then int scan scratch
    const int batch = blockIdx.x >> 1, side = blockIdx.x & 1;
    const int tid = threadIdx.x;
    const int64_t t0 = (int64_t)batch * B;
    const int nb = (int)((n_total - t0) < B ? (n_total - t0) : B); // triples in this batch
    const int n = side ? 2 * nb : nb;
    int n_pad = 64;
    while (n_pad < n)
        n_pad <<= 1;
    int *scratch = (int *)(keys + n_pad);

    for (int k = tid; k < n_pad; k += PREP_THREADS) {
        unsigned long long key = ~0ull;
        if (k < n) {
            if (side == 0) {
                key = ((unsigned long long)(uint32_t)u[t0 + k] << 32) | (uint32_t)k;
            } else { // i-slices first, then j-slices: the side flag is the high payload bit
                int pos = k < nb ? k : k - nb;
                uint32_t sd = k < nb ? 0u : 1u;
                int32_t row = sd ? j[t0 + pos] : i[t0 + pos];
                key = ((unsigned long long)(uint32_t)row << 32) | (sd << 31) | (uint32_t)pos;
            }
        }
        keys[k] = key;
    }
    __syncthreads();

    // bitonic sort, ascending
    for (int k = 2; k <= n_pad; k <<= 1) {
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int t = tid; t < (n_pad >> 1); t += PREP_THREADS) {
                int lo = ((t & ~(jj - 1)) << 1) | (t & (jj - 1));
                int hi = lo + jj;
                unsigned long long a = keys[lo], b = keys[hi];
                bool up = (lo & k) == 0;
                if ((a > b) == up) {
                    keys[lo] = b;
                    keys[hi] = a;
                }
            }
            __syncthreads();
        }
    }

    // heads of equal-row runs -> rank of each unique row
    const int per = n_pad / PREP_THREADS > 0 ? n_pad / PREP_THREADS : 1;
    const int k0 = tid * per;
    int heads = 0;
    for (int k = k0; k < k0 + per && k < n; ++k) {
        uint32_t row = (uint32_t)(keys[k] >> 32);
        heads += (k == 0) || (row != (uint32_t)(keys[k - 1] >> 32));
    }
    int total = 0;
    int rank = block_excl_scan(heads, scratch, &total);

    int32_t *rows_out = side ? w.q_rows + (int64_t)batch * 2 * B : w.u_rows + (int64_t)batch * B;
    int32_t *seg_out = side ? w.q_seg + (int64_t)batch * (2 * B + 1) : w.u_seg + (int64_t)batch * (B + 1);
    int4 *occ_out = side ? w.q_occ + (int64_t)batch * 2 * B : w.u_occ + (int64_t)batch * B;
    for (int k = k0; k < k0 + per && k < n; ++k) {
        unsigned long long key = keys[k];
        uint32_t row = (uint32_t)(key >> 32);
        if (k == 0 || row != (uint32_t)(keys[k - 1] >> 32)) {
            rows_out[rank] = (int32_t)row;
            seg_out[rank] = k;
            ++rank;
        }
        const uint32_t payload = (uint32_t)key;
        const int pos = (int)(payload & 0x7fffffffu);
        occ_out[k] = make_int4(u[t0 + pos], i[t0 + pos], j[t0 + pos], (int)payload);
    }
    if (tid == 0) {
        seg_out[total] = n;
        (side ? w.n_q : w.n_u)[batch] = total;
    }

    // tf1 mode: every table row's {segment start, length} by two binary searches in the sorted keys
    const int64_t n_rows = side ? w.I : w.U;
    if (n_rows > 0) {
        int2 *rs = side ? w.q_rowseg + (int64_t)batch * w.I : w.u_rowseg + (int64_t)batch * w.U;
        for (int64_t r = tid; r < n_rows; r += PREP_THREADS) {
            const int a = lower_bound_key(keys, n, (unsigned long long)r << 32);
            const int b = lower_bound_key(keys, n, (unsigned long long)(r + 1) << 32);
            rs[r] = make_int2(a, b - a);
        }
    }
}

// ------------------------------------------------------------------------------------
// shared device pieces
// ------------------------------------------------------------------------------------
struct Hyper {
    float lr_t, beta1, beta2, eps, reg;
};

struct Tables { // one buffer set
    float *P, *mP, *vP, *Q, *mQ, *vQ;
};

__device__ inline float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ inline void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ inline float dot4(float4 a, float4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }

template <int CTRL> __device__ inline float dpp_add(float v)
{
    // v + (v read through the DPP lane pattern CTRL); all lanes active
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// all-reduce over the LPR lanes of a group; every lane ends with the bit-identical sum.
// xor-1 / xor-2 inside quads, then mirror patterns (a quad, then an 8-lane half, already holds one
// value, so a mirror is as good as an xor), then real cross-row shuffles above 16 lanes.
template <int LPR> __device__ inline float group_sum(float v)
{
    if (LPR >= 2)
        v = dpp_add<0xB1>(v); // quad_perm [1,0,3,2]
    if (LPR >= 4)
        v = dpp_add<0x4E>(v); // quad_perm [2,3,0,1]
    if (LPR >= 8)
        v = dpp_add<0x141>(v); // row_half_mirror
    if (LPR >= 16)
        v = dpp_add<0x140>(v); // row_mirror
    if (LPR >= 32)
        v += __shfl_xor(v, 16, 64);
    if (LPR >= 64)
        v += __shfl_xor(v, 32, 64);
    return v;
}

// sum over the G groups of a wave (same lane position in each group)
template <int LPR> __device__ inline float4 cross_group_sum(float4 g)
{
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        if (off == 1) {
            g.x = dpp_add<0xB1>(g.x), g.y = dpp_add<0xB1>(g.y), g.z = dpp_add<0xB1>(g.z), g.w = dpp_add<0xB1>(g.w);
        } else if (off == 2) {
            g.x = dpp_add<0x4E>(g.x), g.y = dpp_add<0x4E>(g.y), g.z = dpp_add<0x4E>(g.z), g.w = dpp_add<0x4E>(g.w);
        } else {
            g.x += __shfl_xor(g.x, off, 64);
            g.y += __shfl_xor(g.y, off, 64);
            g.z += __shfl_xor(g.z, off, 64);
            g.w += __shfl_xor(g.w, off, 64);
        }
    }
    return g;
}

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

// TF1 Adam on one float4 of a row.  touched == false: the g = 0 sweep of a row the batch
// does not index (m, v still decay and var still moves: _apply_sparse_shared).
__device__ inline void adam4(float4 &var, float4 &m, float4 &v, float4 g, bool touched, const Hyper &h)
{
    const float omb1 = 1.0f - h.beta1, omb2 = 1.0f - h.beta2;
    float *pv = &var.x, *pm = &m.x, *pw = &v.x;
    const float *pg = &g.x;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float mk = pm[c] * h.beta1;
        float vk = pw[c] * h.beta2;
        if (touched) {
            mk = mk + pg[c] * omb1;
            vk = vk + (pg[c] * pg[c]) * omb2;
        }
        pm[c] = mk;
        pw[c] = vk;
        pv[c] = pv[c] - (h.lr_t * mk) / (sqrtf(vk) + h.eps);
    }
}

__device__ inline int4 shfl4(int4 v, int src)
{
    return make_int4(__shfl(v.x, src, 64), __shfl(v.y, src, 64), __shfl(v.z, src, 64), __shfl(v.w, src, 64));
}

// x_b and s_b of one triple from the group's float4 slices: the ONE place this arithmetic lives,
// so user-row and item-row waves agree bit for bit.
template <int LPR> __device__ inline float triple_s(float4 p, float4 qi, float4 qj, float *x_out)
{
    const float di = group_sum<LPR>(dot4(p, qi));
    const float dj = group_sum<LPR>(dot4(p, qj));
    const float x = di - dj; // ui_scores - uj_scores (BPR.py:39-42)
    *x_out = x;
    return 1.0f / (1.0f + expf(x)); // sigmoid(-x) = -d softplus(-x)/dx
}

// ------------------------------------------------------------------------------------
// one user row: sum over its triples of dL/dp, loss terms, (lazy mode) contribution rows
// ------------------------------------------------------------------------------------
template <int LPR, bool STORE_CONTRIB>
__device__ inline float4 user_row_grad(float4 p, const float *__restrict__ Q, int d, const int4 *__restrict__ occ,
                                       int s0, int n, float *__restrict__ contrib, float *__restrict__ loss_triple,
                                       const Hyper &h)
{
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const bool act = 4 * l < d;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float np = group_sum<LPR>(dot4(p, p));
    float4 acc = zero;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int nc = (n - c0) < 64 ? (n - c0) : 64;
        const int4 mine = lane < nc ? occ[s0 + c0 + lane] : make_int4(0, 0, 0, 0);
        const int n_sub = (nc + G - 1) / G;
        for (int t0 = 0; t0 < n_sub; t0 += UNR) {
            int4 rec[UNR];
            float4 qi[UNR], qj[UNR];
            bool has[UNR];
#pragma unroll
            for (int t = 0; t < UNR; ++t) {
                const int k = (t0 + t) * G + g;
                rec[t] = shfl4(mine, k & 63);
                has[t] = (t0 + t) < n_sub && k < nc;
                qi[t] = qj[t] = zero;
                if (has[t] && act) {
                    qi[t] = ld4(Q + (int64_t)rec[t].y * d + 4 * l);
                    qj[t] = ld4(Q + (int64_t)rec[t].z * d + 4 * l);
                }
            }
#pragma unroll
            for (int t = 0; t < UNR; ++t) {
                if ((t0 + t) >= n_sub)
                    break; // wave-uniform
                float x;
                const float s = triple_s<LPR>(p, qi[t], qj[t], &x);
                const float ni = group_sum<LPR>(dot4(qi[t], qi[t]));
                const float nj = group_sum<LPR>(dot4(qj[t], qj[t]));
                if (has[t]) {
                    const int pos = rec[t].w & 0x7fffffff;
                    if (l == 0)
                        loss_triple[pos] = tf_softplus(-x) + h.reg * ((0.5f * np + 0.5f * ni) + 0.5f * nj);
                    if (act) {
                        if (STORE_CONTRIB)
                            st4(contrib + (int64_t)pos * d + 4 * l, make_float4(s * p.x, s * p.y, s * p.z, s * p.w));
                        const float ns = -s;
                        acc.x = acc.x + (ns * (qi[t].x - qj[t].x) + h.reg * p.x);
                        acc.y = acc.y + (ns * (qi[t].y - qj[t].y) + h.reg * p.y);
                        acc.z = acc.z + (ns * (qi[t].z - qj[t].z) + h.reg * p.z);
                        acc.w = acc.w + (ns * (qi[t].w - qj[t].w) + h.reg * p.w);
                    }
                }
            }
        }
    }
    return cross_group_sum<LPR>(acc);
}

// ------------------------------------------------------------------------------------
// one item row: sum over its occurrences of dL/dq.
//   FROM_CONTRIB: c_b is read back (lazy mode).  Otherwise s_b is recomputed from the
//   read-side tables (fused tf1 step).
// ------------------------------------------------------------------------------------
template <int LPR, bool FROM_CONTRIB>
__device__ inline float4 item_row_grad(float4 q, int32_t row, const float *__restrict__ P, const float *__restrict__ Q,
                                       int d, const int4 *__restrict__ occ, int s0, int n,
                                       const float *__restrict__ contrib, const Hyper &h)
{
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const bool act = 4 * l < d;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc = zero;
    (void)row;
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int nc = (n - c0) < 64 ? (n - c0) : 64;
        const int4 mine = lane < nc ? occ[s0 + c0 + lane] : make_int4(0, 0, 0, 0);
        const int n_sub = (nc + G - 1) / G;
        for (int t0 = 0; t0 < n_sub; t0 += UNR) {
            int4 rec[UNR];
            float4 a[UNR], b[UNR]; // FROM_CONTRIB: a = c_b.  else a = p_u, b = the triple's other item row
            bool has[UNR];
#pragma unroll
            for (int t = 0; t < UNR; ++t) {
                const int k = (t0 + t) * G + g;
                rec[t] = shfl4(mine, k & 63);
                has[t] = (t0 + t) < n_sub && k < nc;
                a[t] = b[t] = zero;
                if (has[t] && act) {
                    if (FROM_CONTRIB) {
                        a[t] = ld4(contrib + (int64_t)(rec[t].w & 0x7fffffff) * d + 4 * l);
                    } else {
                        a[t] = ld4(P + (int64_t)rec[t].x * d + 4 * l);
                        const int other = rec[t].w < 0 ? rec[t].y : rec[t].z;
                        b[t] = ld4(Q + (int64_t)other * d + 4 * l);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < UNR; ++t) {
                if ((t0 + t) >= n_sub)
                    break; // wave-uniform
                float4 c = a[t];
                const bool is_j = rec[t].w < 0;
                if (!FROM_CONTRIB) {
                    // operands selected, not branched: the cross-lane sums below run with every lane active
                    const float4 qi = is_j ? b[t] : q, qj = is_j ? q : b[t];
                    float x;
                    const float s = triple_s<LPR>(a[t], qi, qj, &x);
                    c = make_float4(s * a[t].x, s * a[t].y, s * a[t].z, s * a[t].w);
                }
                if (has[t] && act) {
                    const float sg = is_j ? 1.0f : -1.0f; // j-slices get +s*p, i-slices -s*p
                    acc.x = acc.x + (sg * c.x + h.reg * q.x);
                    acc.y = acc.y + (sg * c.y + h.reg * q.y);
                    acc.z = acc.z + (sg * c.z + h.reg * q.z);
                    acc.w = acc.w + (sg * c.w + h.reg * q.w);
                }
            }
        }
    }
    return cross_group_sum<LPR>(acc);
}

// ------------------------------------------------------------------------------------
// tf1 mode: the whole step in one launch, tables read from `src`, written to `dst`
// ------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void bpr_step_fused(Tables src, Tables dst, int d, int64_t U, int64_t I,
                                                      const int2 *__restrict__ u_rowseg,
                                                      const int4 *__restrict__ u_occ,
                                                      const int2 *__restrict__ q_rowseg,
                                                      const int4 *__restrict__ q_occ, float *__restrict__ loss_triple,
                                                      Hyper h)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave_id >= U + I)
        return;
    const bool is_user = wave_id < U;
    const int64_t r = is_user ? wave_id : wave_id - U;
    const int g = lane / LPR, l = lane % LPR;
    const bool act = 4 * l < d;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t ro = r * d + 4 * l;
    const int2 rs = is_user ? u_rowseg[r] : q_rowseg[r];
    const float *sv = is_user ? src.P : src.Q, *sm = is_user ? src.mP : src.mQ, *sw = is_user ? src.vP : src.vQ;
    float4 var = act ? ld4(sv + ro) : zero;
    float4 m = zero, v = zero;
    if (g == 0 && act) {
        m = ld4(sm + ro);
        v = ld4(sw + ro);
    }
    float4 grad = zero;
    if (rs.y > 0) { // wave-uniform
        if (is_user)
            grad = user_row_grad<LPR, false>(var, src.Q, d, u_occ, rs.x, rs.y, nullptr, loss_triple, h);
        else
            grad = item_row_grad<LPR, false>(var, (int32_t)r, src.P, src.Q, d, q_occ, rs.x, rs.y, nullptr, h);
    }
    if (g == 0 && act) {
        adam4(var, m, v, grad, rs.y > 0, h);
        float *dv = is_user ? dst.P : dst.Q, *dm = is_user ? dst.mP : dst.mQ, *dw = is_user ? dst.vP : dst.vQ;
        st4(dv + ro, var);
        st4(dm + ro, m);
        st4(dw + ro, v);
    }
}

// ------------------------------------------------------------------------------------
// lazy mode: touched rows only, in place, two launches
// ------------------------------------------------------------------------------------
template <int LPR>
__global__ __launch_bounds__(256) void bpr_user_side(Tables t, int d, const int32_t *__restrict__ n_rows,
                                                     const int32_t *__restrict__ rows, const int32_t *__restrict__ seg,
                                                     const int4 *__restrict__ occ, float *__restrict__ contrib,
                                                     float *__restrict__ loss_triple, Hyper h)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave_id >= *n_rows)
        return;
    const int64_t r = rows[wave_id];
    const int s0 = seg[wave_id], n = seg[wave_id + 1] - s0;
    const int g = lane / LPR, l = lane % LPR;
    const bool act = 4 * l < d;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t ro = r * d + 4 * l;
    float4 p = act ? ld4(t.P + ro) : zero;
    float4 m = zero, v = zero;
    if (g == 0 && act) {
        m = ld4(t.mP + ro);
        v = ld4(t.vP + ro);
    }
    const float4 grad = user_row_grad<LPR, true>(p, t.Q, d, occ, s0, n, contrib, loss_triple, h);
    if (g == 0 && act) {
        adam4(p, m, v, grad, true, h);
        st4(t.P + ro, p);
        st4(t.mP + ro, m);
        st4(t.vP + ro, v);
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void bpr_item_side(Tables t, int d, const int32_t *__restrict__ n_rows,
                                                     const int32_t *__restrict__ rows, const int32_t *__restrict__ seg,
                                                     const int4 *__restrict__ occ, const float *__restrict__ contrib,
                                                     Hyper h)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wave_id >= *n_rows)
        return;
    const int64_t r = rows[wave_id];
    const int s0 = seg[wave_id], n = seg[wave_id + 1] - s0;
    const int g = lane / LPR, l = lane % LPR;
    const bool act = 4 * l < d;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t ro = r * d + 4 * l;
    float4 q = act ? ld4(t.Q + ro) : zero;
    float4 m = zero, v = zero;
    if (g == 0 && act) {
        m = ld4(t.mQ + ro);
        v = ld4(t.vQ + ro);
    }
    const float4 grad = item_row_grad<LPR, true>(q, (int32_t)r, nullptr, nullptr, d, occ, s0, n, contrib, h);
    if (g == 0 && act) {
        adam4(q, m, v, grad, true, h);
        st4(t.Q + ro, q);
        st4(t.mQ + ro, m);
        st4(t.vQ + ro, v);
    }
}

// ------------------------------------------------------------------------------------
// bpr_reduce_loss: one workgroup per batch, fixed summation order, float64 accumulation
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bpr_reduce_loss(const float *__restrict__ loss_triple, int64_t n_total,
                                                       int32_t B, float *__restrict__ loss_per_batch)
{
    __shared__ double part[256];
    const int64_t t0 = (int64_t)blockIdx.x * B;
    const int nb = (int)((n_total - t0) < B ? (n_total - t0) : B);
    double s = 0.0;
    for (int k = threadIdx.x; k < nb; k += 256)
        s += (double)loss_triple[t0 + k];
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

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
struct StepArgs {
    crec_ctx *ctx;
    Tables cur, other; // tf1: read `cur`, write `other`.  lazy: `cur` in place.
    int d;
    int64_t U, I;
    int32_t B;
    float *contrib, *loss_triple;
    Hyper h;
    hipStream_t st;
};

template <int LPR> void launch_fused(const StepArgs &a, const IndexWs &w, int64_t b)
{
    const int64_t waves = a.U + a.I;
    crec_prof_begin(a.ctx, CREC_K_STEP_FUSED, a.st);
    hipLaunchKernelGGL((bpr_step_fused<LPR>), dim3((unsigned)crec_ceil_div(waves, 4)), dim3(256), 0, a.st, a.cur, a.other,
                       a.d, a.U, a.I, w.u_rowseg + b * w.U, w.u_occ + b * w.B, w.q_rowseg + b * w.I,
                       w.q_occ + b * 2 * w.B, a.loss_triple, a.h);
    crec_prof_end(a.ctx, a.st);
}

template <int LPR> void launch_lazy(const StepArgs &a, const IndexWs &w, int64_t b)
{
    const int64_t u_waves = a.U < a.B ? a.U : a.B;
    const int64_t q_waves = a.I < 2 * (int64_t)a.B ? a.I : 2 * (int64_t)a.B;
    crec_prof_begin(a.ctx, CREC_K_USER_SIDE, a.st);
    hipLaunchKernelGGL((bpr_user_side<LPR>), dim3((unsigned)crec_ceil_div(u_waves, 4)), dim3(256), 0, a.st, a.cur, a.d,
                       w.n_u + b, w.u_rows + b * w.B, w.u_seg + b * (w.B + 1), w.u_occ + b * w.B, a.contrib,
                       a.loss_triple, a.h);
    crec_prof_end(a.ctx, a.st);
    crec_prof_begin(a.ctx, CREC_K_ITEM_SIDE, a.st);
    hipLaunchKernelGGL((bpr_item_side<LPR>), dim3((unsigned)crec_ceil_div(q_waves, 4)), dim3(256), 0, a.st, a.cur, a.d,
                       w.n_q + b, w.q_rows + b * 2 * w.B, w.q_seg + b * (2 * w.B + 1), w.q_occ + b * 2 * w.B,
                       (const float *)a.contrib, a.h);
    crec_prof_end(a.ctx, a.st);
}

using StepFn = void (*)(const StepArgs &, const IndexWs &, int64_t);

StepFn pick_step(int d, bool dense)
{
    int lpr = 1;
    while (lpr * 4 < d)
        lpr <<= 1;
#define CREC_PICK(L) \
    case L:          \
        return dense ? launch_fused<L> : launch_lazy<L>;
    switch (lpr) {
        CREC_PICK(1)
        CREC_PICK(2)
        CREC_PICK(4)
        CREC_PICK(8)
        CREC_PICK(16)
        CREC_PICK(32)
        CREC_PICK(64)
    }
#undef CREC_PICK
    return nullptr;
}

} // namespace

extern "C" int crec_bpr_train_epoch(crec_ctx *ctx, const crec_bpr_tables *t, crec_adam *opt, const int32_t *u,
                                    const int32_t *i, const int32_t *j, int64_t N, int32_t B, float *loss_per_batch,
                                    void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, t && opt, "null argument");
    CREC_REQUIRE(ctx, N >= 0, "negative N");
    CREC_REQUIRE(ctx, N == 0 || (u && i && j && loss_per_batch), "null argument");
    CREC_REQUIRE(ctx, t->P && t->Q && t->mP && t->vP && t->mQ && t->vQ, "null table pointer");
    CREC_REQUIRE(ctx, t->d >= 4 && t->d <= 256 && t->d % 4 == 0, "embed_size %d: need a multiple of 4 in [4, 256]", t->d);
    CREC_REQUIRE(ctx, t->U >= 1 && t->I >= 1 && t->U < ((int64_t)1 << 31) && t->I < ((int64_t)1 << 31), "table rows");
    CREC_REQUIRE(ctx, B >= 1 && B <= MAX_B, "batch_size %d: this build indexes batches of at most %d in LDS", B, MAX_B);
    CREC_REQUIRE(ctx, opt->adam_mode == CREC_ADAM_TF1 || opt->adam_mode == CREC_ADAM_LAZY, "adam_mode");
    const float *tabs[] = {t->P, t->Q, t->mP, t->vP, t->mQ, t->vQ};
    for (const float *p : tabs)
        CREC_REQUIRE(ctx, ((uintptr_t)p & 15) == 0, "table base not 16-byte aligned");
    if (N == 0)
        return CREC_OK;
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const bool dense = opt->adam_mode == CREC_ADAM_TF1;
    StepFn step = pick_step(t->d, dense);
    const int64_t nb = crec_ceil_div(N, B);

    // chunk the epoch so the inverted indices of a chunk stay under ~1 GiB
    const size_t per_batch = layout_index(nullptr, 1, B, t->U, t->I, dense).bytes;
    int64_t chunk = (int64_t)((size_t)1 << 30) / (int64_t)per_batch;
    chunk = chunk < 1 ? 1 : (chunk > nb ? nb : chunk);
    if (int rc = crec_reserve(ctx, &ctx->idx_ws, layout_index(nullptr, chunk, B, t->U, t->I, dense).bytes))
        return rc;
    if (int rc = crec_reserve(ctx, &ctx->loss_ws, sizeof(float) * (size_t)chunk * B))
        return rc;

    StepArgs a{};
    a.ctx = ctx;
    a.cur = Tables{t->P, t->mP, t->vP, t->Q, t->mQ, t->vQ};
    a.d = t->d;
    a.U = t->U;
    a.I = t->I;
    a.B = B;
    a.st = st;
    const size_t bytesP = sizeof(float) * (size_t)t->U * t->d, bytesQ = sizeof(float) * (size_t)t->I * t->d;
    if (dense) { // the second buffer set of the ping-pong
        const size_t aP = align_up(bytesP, 256), aQ = align_up(bytesQ, 256);
        if (int rc = crec_reserve(ctx, &ctx->shadow, 3 * aP + 3 * aQ))
            return rc;
        char *s = (char *)ctx->shadow.ptr;
        a.other = Tables{(float *)s, (float *)(s + aP), (float *)(s + 2 * aP), (float *)(s + 3 * aP),
                         (float *)(s + 3 * aP + aQ), (float *)(s + 3 * aP + 2 * aQ)};
    } else {
        if (int rc = crec_reserve(ctx, &ctx->contrib, sizeof(float) * (size_t)B * t->d))
            return rc;
        a.contrib = (float *)ctx->contrib.ptr;
    }

    // dynamic LDS of the index kernel: padded key array + scan scratch
    int n_pad = 64;
    while (n_pad < 2 * B)
        n_pad <<= 1;
    const size_t lds = sizeof(unsigned long long) * n_pad + sizeof(int) * (PREP_THREADS / 64 + 2);
    CREC_HIP(ctx, hipFuncSetAttribute((const void *)bpr_index_batches, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));

    bool in_shadow = false; // tf1: which buffer set holds the current tables
    for (int64_t b0 = 0; b0 < nb; b0 += chunk) {
        const int64_t nbc = (nb - b0) < chunk ? (nb - b0) : chunk;
        const int64_t t0 = b0 * B, n_chunk = (N - t0) < nbc * B ? (N - t0) : nbc * B;
        IndexWs w = layout_index(ctx->idx_ws.ptr, nbc, B, t->U, t->I, dense);
        crec_prof_begin(ctx, CREC_K_INDEX, st);
        hipLaunchKernelGGL(bpr_index_batches, dim3((unsigned)(2 * nbc)), dim3(PREP_THREADS), lds, st, u + t0, i + t0,
                           j + t0, n_chunk, B, w);
        crec_prof_end(ctx, st);
        for (int64_t b = 0; b < nbc; ++b) {
            // lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t) with fp32 running powers (TF1 _prepare/_finish)
            a.h.lr_t = (opt->lr * sqrtf(1.0f - opt->beta2_power)) / (1.0f - opt->beta1_power);
            a.h.beta1 = opt->beta1;
            a.h.beta2 = opt->beta2;
            a.h.eps = opt->eps;
            a.h.reg = opt->reg;
            a.loss_triple = (float *)ctx->loss_ws.ptr + b * B;
            step(a, w, b);
            if (dense) {
                Tables tmp = a.cur;
                a.cur = a.other;
                a.other = tmp;
                in_shadow = !in_shadow;
            }
            opt->beta1_power = opt->beta1_power * opt->beta1;
            opt->beta2_power = opt->beta2_power * opt->beta2;
            opt->step += 1;
        }
        crec_prof_begin(ctx, CREC_K_REDUCE_LOSS, st);
        hipLaunchKernelGGL(bpr_reduce_loss, dim3((unsigned)nbc), dim3(256), 0, st, (const float *)ctx->loss_ws.ptr,
                           n_chunk, B, loss_per_batch + b0);
        crec_prof_end(ctx, st);
        CREC_HIP(ctx, hipGetLastError());
        if (int rc = crec_prof_flush(ctx, st))
            return rc;
    }
    if (in_shadow) { // an odd number of steps: bring the result home to the caller's buffers
        const Tables &s = a.cur;
        CREC_HIP(ctx, hipMemcpyAsync(t->P, s.P, bytesP, hipMemcpyDeviceToDevice, st));
        CREC_HIP(ctx, hipMemcpyAsync(t->mP, s.mP, bytesP, hipMemcpyDeviceToDevice, st));
        CREC_HIP(ctx, hipMemcpyAsync(t->vP, s.vP, bytesP, hipMemcpyDeviceToDevice, st));
        CREC_HIP(ctx, hipMemcpyAsync(t->Q, s.Q, bytesQ, hipMemcpyDeviceToDevice, st));
        CREC_HIP(ctx, hipMemcpyAsync(t->mQ, s.mQ, bytesQ, hipMemcpyDeviceToDevice, st));
        CREC_HIP(ctx, hipMemcpyAsync(t->vQ, s.vQ, bytesQ, hipMemcpyDeviceToDevice, st));
    }
    return CREC_OK;
}
