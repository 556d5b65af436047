// BPR training step for gfx950: gather -> pairwise loss -> per-row gradient sums -> Adam,
// without atomics and without ever materialising gradient rows in HBM.
//
// Replaces, per batch, sess.run([train, loss]) on the graph of model/ranking/BPR.py:31-44
// with tf.train.AdamOptimizer semantics (utils/tools.py:83-84); see include/crec.h.
//
// Per chunk of batches, one launch:
//   bpr_index_batches  one workgroup per (batch, side): bitonic sort in LDS of (row, position)
//                      keys -> unique rows, segment starts, occurrence records.  This is the
//                      "unique + unsorted_segment_sum" bookkeeping of TF1's sparse Adam, done
//                      once for the whole chunk and in batch position order, i-slices before
//                      j-slices, so row sums are deterministic.
//   bpr_row_words      (tf1 mode) per 32 table rows: {touched bitmap, rank of first} so the
//                      dense sweep finds a row's segment with one 8-byte load.
// Per batch, two launches (a step's row updates must not be seen by that step's gathers):
//   bpr_user_side      one wave per user row: for every triple of that user gather q_i, q_j,
//                      x = p.q_i - p.q_j, s = sigmoid(-x), loss term, contribution row
//                      c_b = s*p_u (plain coalesced store), sum of dL/dp; then Adam on the row
//                      in place.  P rows are read only by their own wave, so in place is safe.
//   bpr_item_side      one wave per item row: sum of -/+ c_b over the row's i-/j-occurrences
//                      plus reg*q; Adam in place.  Reads no other Q row.
// Per chunk: bpr_reduce_loss sums the per-triple loss terms of each batch in a fixed order.
//
// A wave splits into G = 64/LPR groups of LPR lanes, each lane holding a float4 of a row
// (LPR*4 >= d), so G occurrences are in flight per wave and every row access is one
// coalesced 16-byte-per-lane load.  Dots reduce with xor-shuffles inside a group.
//
// Compiled with -ffp-contract=off: each fp32 multiply/add is rounded separately, like the
// element-wise Eigen kernels TF1 runs (and like the CPU oracle), so the Adam arithmetic is
// bit-reproducible.
#include "crec_internal.h"

namespace {

constexpr int PREP_THREADS = 1024;
constexpr int MAX_B = 8192;

// ------------------------------------------------------------------------------------
// index workspace layout for one chunk of nbc batches
// ------------------------------------------------------------------------------------
struct IndexWs {
    int32_t *n_u, *n_q;       // [nbc] unique rows per batch
    int32_t *u_rows, *u_seg;  // [nbc][B], [nbc][B+1]
    int4 *u_occ;              // [nbc][B]   {i, j, position in batch, user row}
    int32_t *q_rows, *q_seg;  // [nbc][2B], [nbc][2B+1]
    int32_t *q_occ;           // [nbc][2B]  position | sign<<31  (sign 1 = the triple's j)
    int2 *u_words, *q_words;  // [nbc][WU], [nbc][WI]  {bitmap of 32 rows, rank of the first}
    int64_t B, WU, WI;
    size_t bytes;
};

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

IndexWs layout_index(void *base, int64_t nbc, int64_t B, int64_t U, int64_t I, bool dense)
{
    IndexWs w{};
    w.B = B;
    w.WU = dense ? crec_ceil_div(U, 32) : 0;
    w.WI = dense ? crec_ceil_div(I, 32) : 0;
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
    w.q_occ = (int32_t *)take(sizeof(int32_t) * nbc * 2 * B);
    w.u_words = (int2 *)take(sizeof(int2) * nbc * w.WU);
    w.q_words = (int2 *)take(sizeof(int2) * nbc * w.WI);
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
            } else { // i-slices first, then j-slices: sign is the high payload bit
                int pos = k < nb ? k : k - nb;
                uint32_t sign = k < nb ? 0u : 1u;
                int32_t row = sign ? j[t0 + pos] : i[t0 + pos];
                key = ((unsigned long long)(uint32_t)row << 32) | (sign << 31) | (uint32_t)pos;
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
    for (int k = k0; k < k0 + per && k < n; ++k) {
        unsigned long long key = keys[k];
        uint32_t row = (uint32_t)(key >> 32);
        if (k == 0 || row != (uint32_t)(keys[k - 1] >> 32)) {
            rows_out[rank] = (int32_t)row;
            seg_out[rank] = k;
            ++rank;
        }
        uint32_t payload = (uint32_t)key;
        if (side == 0) {
            int pos = (int)payload;
            w.u_occ[(int64_t)batch * B + k] = make_int4(i[t0 + pos], j[t0 + pos], pos, (int32_t)row);
        } else {
            w.q_occ[(int64_t)batch * 2 * B + k] = (int32_t)payload;
        }
    }
    if (tid == 0) {
        seg_out[total] = n;
        (side ? w.n_q : w.n_u)[batch] = total;
    }
}

// ------------------------------------------------------------------------------------
// bpr_row_words (tf1 mode): grid = 2*nbc, one workgroup per (batch, side)
// ------------------------------------------------------------------------------------
__global__ void bpr_row_words(IndexWs w, int64_t U, int64_t I)
{
    const int batch = blockIdx.x >> 1, side = blockIdx.x & 1;
    const int32_t *rows = side ? w.q_rows + (int64_t)batch * 2 * w.B : w.u_rows + (int64_t)batch * w.B;
    const int n = (side ? w.n_q : w.n_u)[batch];
    const int64_t W = side ? w.WI : w.WU;
    int2 *words = side ? w.q_words + (int64_t)batch * W : w.u_words + (int64_t)batch * W;
    (void)U;
    (void)I;
    for (int64_t wd = threadIdx.x; wd < W; wd += blockDim.x) {
        const int64_t first = wd * 32;
        int lo = 0, hi = n; // lower_bound(rows, first)
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (rows[mid] < first)
                lo = mid + 1;
            else
                hi = mid;
        }
        uint32_t bits = 0;
        for (int k = lo; k < n && rows[k] < first + 32; ++k)
            bits |= 1u << (rows[k] - first);
        words[wd] = make_int2((int)bits, lo);
    }
}

// ------------------------------------------------------------------------------------
// shared pieces of the two row kernels
// ------------------------------------------------------------------------------------
struct Hyper {
    float lr_t, beta1, beta2, eps, reg;
};

__device__ inline float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ inline void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ inline float dot4(float4 a, float4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }

template <int LPR> __device__ inline float group_sum(float v)
{
#pragma unroll
    for (int off = LPR >> 1; off > 0; off >>= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

template <int LPR> __device__ inline float4 cross_group_sum(float4 g)
{
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        g.x += __shfl_xor(g.x, off, 64);
        g.y += __shfl_xor(g.y, off, 64);
        g.z += __shfl_xor(g.z, off, 64);
        g.w += __shfl_xor(g.w, off, 64);
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

// wave -> (row, segment) for both modes
template <bool DENSE>
__device__ inline bool locate_row(int64_t wave_id, int64_t n_table_rows, const int32_t *n_rows, const int32_t *rows,
                                  const int32_t *seg, const int2 *words, int64_t *row, int *s0, int *s1)
{
    if (DENSE) {
        if (wave_id >= n_table_rows)
            return false;
        *row = wave_id;
        int2 wd = words[wave_id >> 5];
        uint32_t bits = (uint32_t)wd.x, bit = 1u << (wave_id & 31);
        if (bits & bit) {
            int rank = wd.y + __popc(bits & (bit - 1));
            *s0 = seg[rank];
            *s1 = seg[rank + 1];
        } else {
            *s0 = *s1 = 0;
        }
        return true;
    }
    if (wave_id >= *n_rows)
        return false;
    *row = rows[wave_id];
    *s0 = seg[wave_id];
    *s1 = seg[wave_id + 1];
    return true;
}

// ------------------------------------------------------------------------------------
// bpr_user_side
// ------------------------------------------------------------------------------------
template <int LPR, bool DENSE>
__global__ __launch_bounds__(256) void bpr_user_side(float *__restrict__ P, float *__restrict__ mP,
                                                     float *__restrict__ vP, const float *__restrict__ Q, int d,
                                                     int64_t U, const int32_t *__restrict__ n_rows,
                                                     const int32_t *__restrict__ rows, const int32_t *__restrict__ seg,
                                                     const int4 *__restrict__ occ, const int2 *__restrict__ words,
                                                     float *__restrict__ contrib, float *__restrict__ loss_triple,
                                                     Hyper h)
{
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int64_t r;
    int s0, s1;
    if (!locate_row<DENSE>(wave_id, U, n_rows, rows, seg, words, &r, &s0, &s1))
        return;
    const int g = lane / LPR, l = lane % LPR;
    const bool act = 4 * l < d;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t ro = r * d + 4 * l;
    float4 p = act ? ld4(P + ro) : zero;
    float4 m = zero, v = zero;
    if (g == 0 && act) {
        m = ld4(mP + ro);
        v = ld4(vP + ro);
    }
    const float np = group_sum<LPR>(dot4(p, p));
    float4 acc = zero;
    for (int base = s0; base < s1; base += G) {
        const int k = base + g;
        const bool has = k < s1;
        int4 rec = has ? occ[k] : make_int4(0, 0, 0, 0);
        float4 qi = zero, qj = zero;
        if (has && act) {
            qi = ld4(Q + (int64_t)rec.x * d + 4 * l);
            qj = ld4(Q + (int64_t)rec.y * d + 4 * l);
        }
        const float di = group_sum<LPR>(dot4(p, qi));
        const float dj = group_sum<LPR>(dot4(p, qj));
        const float ni = group_sum<LPR>(dot4(qi, qi));
        const float nj = group_sum<LPR>(dot4(qj, qj));
        const float x = di - dj;
        const float s = 1.0f / (1.0f + expf(x)); // sigmoid(-x)
        if (has) {
            if (l == 0)
                loss_triple[rec.z] = tf_softplus(-x) + h.reg * ((0.5f * np + 0.5f * ni) + 0.5f * nj);
            if (act) {
                st4(contrib + (int64_t)rec.z * d + 4 * l, make_float4(s * p.x, s * p.y, s * p.z, s * p.w));
                const float ns = -s;
                acc.x = acc.x + (ns * (qi.x - qj.x) + h.reg * p.x);
                acc.y = acc.y + (ns * (qi.y - qj.y) + h.reg * p.y);
                acc.z = acc.z + (ns * (qi.z - qj.z) + h.reg * p.z);
                acc.w = acc.w + (ns * (qi.w - qj.w) + h.reg * p.w);
            }
        }
    }
    acc = cross_group_sum<LPR>(acc);
    if (g == 0 && act) {
        adam4(p, m, v, acc, s1 > s0, h);
        st4(P + ro, p);
        st4(mP + ro, m);
        st4(vP + ro, v);
    }
}

// ------------------------------------------------------------------------------------
// bpr_item_side
// ------------------------------------------------------------------------------------
template <int LPR, bool DENSE>
__global__ __launch_bounds__(256) void bpr_item_side(float *__restrict__ Q, float *__restrict__ mQ,
                                                     float *__restrict__ vQ, int d, int64_t I,
                                                     const int32_t *__restrict__ n_rows,
                                                     const int32_t *__restrict__ rows, const int32_t *__restrict__ seg,
                                                     const int32_t *__restrict__ occ, const int2 *__restrict__ words,
                                                     const float *__restrict__ contrib, Hyper h)
{
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    int64_t r;
    int s0, s1;
    if (!locate_row<DENSE>(wave_id, I, n_rows, rows, seg, words, &r, &s0, &s1))
        return;
    const int g = lane / LPR, l = lane % LPR;
    const bool act = 4 * l < d;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t ro = r * d + 4 * l;
    float4 q = act ? ld4(Q + ro) : zero;
    float4 m = zero, v = zero;
    if (g == 0 && act) {
        m = ld4(mQ + ro);
        v = ld4(vQ + ro);
    }
    float4 acc = zero;
    for (int base = s0; base < s1; base += G) {
        const int k = base + g;
        if (k < s1 && act) {
            const int32_t rec = occ[k];
            const int pos = rec & 0x7fffffff;
            float4 c = ld4(contrib + (int64_t)pos * d + 4 * l);
            const float sg = rec < 0 ? 1.0f : -1.0f; // j-slices get +s*p, i-slices -s*p
            acc.x = acc.x + (sg * c.x + h.reg * q.x);
            acc.y = acc.y + (sg * c.y + h.reg * q.y);
            acc.z = acc.z + (sg * c.z + h.reg * q.z);
            acc.w = acc.w + (sg * c.w + h.reg * q.w);
        }
    }
    acc = cross_group_sum<LPR>(acc);
    if (g == 0 && act) {
        adam4(q, m, v, acc, s1 > s0, h);
        st4(Q + ro, q);
        st4(mQ + ro, m);
        st4(vQ + ro, v);
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
template <int LPR, bool DENSE>
void launch_step(crec_ctx *ctx, const crec_bpr_tables &t, const IndexWs &w, int64_t b, float *contrib,
                 float *loss_triple, const Hyper &h, int32_t B, hipStream_t st)
{
    const int64_t u_waves = DENSE ? t.U : (t.U < B ? t.U : B);
    const int64_t q_waves = DENSE ? t.I : (t.I < 2 * (int64_t)B ? t.I : 2 * (int64_t)B);
    crec_prof_begin(ctx, CREC_K_USER_SIDE, st);
    hipLaunchKernelGGL((bpr_user_side<LPR, DENSE>), dim3((unsigned)crec_ceil_div(u_waves, 4)), dim3(256), 0, st, t.P,
                       t.mP, t.vP, t.Q, t.d, t.U, w.n_u + b, w.u_rows + b * w.B, w.u_seg + b * (w.B + 1),
                       w.u_occ + b * w.B, w.u_words + b * w.WU, contrib, loss_triple, h);
    crec_prof_end(ctx, st);
    crec_prof_begin(ctx, CREC_K_ITEM_SIDE, st);
    hipLaunchKernelGGL((bpr_item_side<LPR, DENSE>), dim3((unsigned)crec_ceil_div(q_waves, 4)), dim3(256), 0, st, t.Q,
                       t.mQ, t.vQ, t.d, t.I, w.n_q + b, w.q_rows + b * 2 * w.B, w.q_seg + b * (2 * w.B + 1),
                       w.q_occ + b * 2 * w.B, w.q_words + b * w.WI, contrib, h);
    crec_prof_end(ctx, st);
}

using StepFn = void (*)(crec_ctx *, const crec_bpr_tables &, const IndexWs &, int64_t, float *, float *, const Hyper &,
                        int32_t, hipStream_t);

StepFn pick_step(int d, bool dense)
{
    int lpr = 1;
    while (lpr * 4 < d)
        lpr <<= 1;
#define CREC_PICK(L) \
    case L:          \
        return dense ? launch_step<L, true> : launch_step<L, false>;
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
    if (int rc = crec_reserve(ctx, &ctx->contrib, sizeof(float) * (size_t)B * t->d))
        return rc;
    if (int rc = crec_reserve(ctx, &ctx->loss_ws, sizeof(float) * (size_t)chunk * B))
        return rc;

    // dynamic LDS of the index kernel: padded key array + scan scratch
    int n_pad = 64;
    while (n_pad < 2 * B)
        n_pad <<= 1;
    const size_t lds = sizeof(unsigned long long) * n_pad + sizeof(int) * (PREP_THREADS / 64 + 2);
    CREC_HIP(ctx, hipFuncSetAttribute((const void *)bpr_index_batches, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));

    for (int64_t b0 = 0; b0 < nb; b0 += chunk) {
        const int64_t nbc = (nb - b0) < chunk ? (nb - b0) : chunk;
        const int64_t t0 = b0 * B, n_chunk = (N - t0) < nbc * B ? (N - t0) : nbc * B;
        IndexWs w = layout_index(ctx->idx_ws.ptr, nbc, B, t->U, t->I, dense);
        crec_prof_begin(ctx, CREC_K_INDEX, st);
        hipLaunchKernelGGL(bpr_index_batches, dim3((unsigned)(2 * nbc)), dim3(PREP_THREADS), lds, st, u + t0, i + t0,
                           j + t0, n_chunk, B, w);
        crec_prof_end(ctx, st);
        if (dense) {
            crec_prof_begin(ctx, CREC_K_WORDS, st);
            hipLaunchKernelGGL(bpr_row_words, dim3((unsigned)(2 * nbc)), dim3(256), 0, st, w, t->U, t->I);
            crec_prof_end(ctx, st);
        }
        for (int64_t b = 0; b < nbc; ++b) {
            Hyper h;
            // lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t) with fp32 running powers (TF1 _prepare/_finish)
            h.lr_t = (opt->lr * sqrtf(1.0f - opt->beta2_power)) / (1.0f - opt->beta1_power);
            h.beta1 = opt->beta1;
            h.beta2 = opt->beta2;
            h.eps = opt->eps;
            h.reg = opt->reg;
            step(ctx, *t, w, b, (float *)ctx->contrib.ptr, (float *)ctx->loss_ws.ptr + b * B, h, B, st);
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
    return CREC_OK;
}
