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
// A step's updates go out as ONE launch (apply_all: up to four tables of one row-width class, an optional
// one-column table, the small dense vector): these steps are bound by launches, not bytes.
#include "crec_internal.h"
#include "rows_adam.h"
#include "block_sort.h"

namespace {

constexpr int IDX_THREADS = 1024;

inline size_t rows_index_lds_bytes(int cap)
{
    return sizeof(uint32_t) * cap + sizeof(uint16_t) * (2 * (size_t)cap + 64) +
           sizeof(int) * (crec_sort::hist_ints<IDX_THREADS>() + crec_sort::scan_ints<IDX_THREADS>());
}

// one workgroup per (batch, chunk): stable sort of the chunk's (row, slot) pairs in LDS (block_sort.h);
// order[k] = slot of the k-th sorted pair (slot ids and positions are relative to the batch); rowseg[r] = {first
// position of row r, number of slots on row r} for every table row
__global__ __launch_bounds__(IDX_THREADS) void rows_index_batches(const int32_t *__restrict__ idx, int64_t n_total,
                                                                  int32_t per_batch, int32_t chunks, int64_t n_rows,
                                                                  int32_t key_bits, int32_t cap,
                                                                  int32_t *__restrict__ order, int2 *__restrict__ rowseg)
{
    extern __shared__ uint32_t rows[]; // [cap]
    uint16_t *tag = (uint16_t *)(rows + cap);
    uint16_t *seg = tag + cap; // [cap + 64]
    int *hist = (int *)(seg + cap + 64);
    int *scan = hist + crec_sort::hist_ints<IDX_THREADS>();
    const int batch = blockIdx.x / chunks, chunk = blockIdx.x % chunks, tid = threadIdx.x;
    const int64_t t0 = (int64_t)batch * per_batch;
    const int nbatch = (int)((n_total - t0) < per_batch ? (n_total - t0) : per_batch);
    const int clen = (per_batch + chunks - 1) / chunks, c0 = chunk * clen < nbatch ? chunk * clen : nbatch;
    const int n = (nbatch - c0) < clen ? (nbatch - c0) : clen;
    for (int k = tid; k < n; k += IDX_THREADS) {
        rows[k] = (uint32_t)idx[t0 + c0 + k];
        tag[k] = (uint16_t)k;
    }
    crec_sort::block_sort_rows<IDX_THREADS>(rows, tag, n, key_bits, hist, scan);
    for (int k = tid; k < n; k += IDX_THREADS)
        order[t0 + c0 + k] = c0 + (int32_t)tag[k];

    // run starts by rank (blocked over positions: the scan runs in sorted order), then the table rows' segments
    const int per = (n + IDX_THREADS - 1) / IDX_THREADS;
    const int k0 = tid * per, k1 = (k0 + per) < n ? (k0 + per) : n;
    int heads = 0;
    for (int k = k0; k < k1; ++k)
        heads += (k == 0 || rows[k] != rows[k - 1]) ? 1 : 0;
    int total = 0;
    int rank = crec_sort::block_excl_scan<IDX_THREADS>(heads, scan, &total);
    for (int k = k0; k < k1; ++k)
        if (k == 0 || rows[k] != rows[k - 1])
            seg[rank++] = (uint16_t)k;
    if (tid == 0)
        seg[total] = (uint16_t)n;
    __syncthreads();
    int2 *rs = rowseg + ((int64_t)batch * chunks + chunk) * n_rows;
    for (int q = tid; q <= total; q += IDX_THREADS)
        crec_sort::rowseg_fill(rs, n_rows, q, total, n, rows, seg, c0, [](int c) { return c; });
}

__device__ inline float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ inline void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
// rows written once per launch and not read again by it (bpr_train.hip, CREC_NT): non-temporal when CREC_ROWS_NT
#ifndef CREC_ROWS_NT
#define CREC_ROWS_NT 1
#endif
typedef float crec_rows_f4v __attribute__((ext_vector_type(4)));
__device__ inline void st4_stream(float *p, float4 v)
{
    if (CREC_ROWS_NT) {
        crec_rows_f4v t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<crec_rows_f4v *>(p));
    } else {
        st4(p, v);
    }
}

// one element under the step's optimizer (get_optimizer, utils/tools.py:79-87).  `dense`: the ApplyAdam form of Adam
// (dense variables), else the IndexedSlices form; SGD: var -= lr*g; Adagrad: accum += g*g, var -= lr*g/sqrt(accum)
// (accum in m; the same arithmetic as the BPR step's opt4, csrc/bpr_train.hip)
__device__ __forceinline__ float opt_elem(const crec_rows_opt &o, bool dense, float g, float var, float &m, float &v)
{
    if (o.opt == CREC_OPT_SGD)
        return var - o.lr_t * g;
    if (o.opt == CREC_OPT_ADAGRAD) {
        m = m + g * g;
#ifdef CREC_EXACT_MATH
        return var - (o.lr_t * g) / sqrtf(m);
#else
        return var - (o.lr_t * g) * __builtin_amdgcn_rsqf(m);
#endif
    }
    const float omb1 = 1.0f - o.beta1, omb2 = 1.0f - o.beta2;
    if (dense) {
        m = m + (g - m) * omb1;
        v = v + (g * g - v) * omb2;
        return var - (m * o.lr_t) / (sqrtf(v) + o.eps);
    }
    m = m * o.beta1 + g * omb1;
    v = v * o.beta2 + (g * g) * omb2;
    return var - (o.lr_t * m) / (sqrtf(v) + o.eps);
}

// a sparse table's rows no slot names: swept by TF1's Adam (g = 0), left alone by lazy Adam, SGD and Adagrad
__device__ __forceinline__ bool untouched_stays(const crec_rows_opt &o)
{
    return o.kind == ROWS_SPARSE_LAZY || (o.kind == ROWS_SPARSE_TF1 && o.opt != 0);
}

#ifdef CREC_ROWS_STAMPS // debug build (tools/rows_stamps.sh): every wave of apply_all records when it began and ended and when its
                        // loads arrived; [wave][8] = {begin, end, slots of the row, 0, marks 0..3}, 100 MHz clock
__device__ unsigned long long rows_stamps[8 * 16384];
extern "C" int crec_debug_rows_stamps(unsigned long long *out, size_t n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rows_stamps), n * sizeof(unsigned long long));
}
__device__ __forceinline__ void rows_mark(int slot, int dep, bool wait) // the time at which `dep` is available to the wave
{
    if (wait)
        asm volatile("" ::"v"(dep));
    const unsigned long long t = wall_clock64();
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0 && w < 16384)
        rows_stamps[8 * w + slot] = t;
}
#define ROWS_MARK(slot, dep) rows_mark(slot, dep, true)
#define ROWS_NOTE(slot, val)                                                                  \
    do {                                                                                      \
        const int w_ = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                   \
        if ((threadIdx.x & 63) == 0 && w_ < 16384)                                            \
            rows_stamps[8 * w_ + slot] = (unsigned long long)(val);                           \
    } while (0)
#else
#define ROWS_MARK(slot, dep)
#define ROWS_NOTE(slot, val)
#endif

// one wave per table row; group g of the wave takes slots g, g+G, ... of the row (fixed order),
// the G partial sums are added in group order
constexpr int SEGS = 2; // row segments (one per chunk of the index) in flight together
template <int LPR, bool CHUNKED = false>
__device__ __forceinline__ void rows_apply_row(const crec_rows_table &t, const int2 *__restrict__ rowseg,
                                               const int32_t *__restrict__ order, const float *__restrict__ contrib,
                                               const float *__restrict__ dense_grad, float *__restrict__ row_out,
                                               const crec_rows_opt &o, int64_t r, int chunks = 1)
{
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    if (r >= t.rows)
        return;
    const int d = t.d;
    const bool act = 4 * l < d;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t ro = r * d + 4 * l;
    const int2 rs0 = rowseg[r];
    ROWS_MARK(4, rs0.y); // the first segment has arrived
    if (chunks == 1 && rs0.y == 0 && untouched_stays(o))
        return;
    const float4 var = act ? ld4(t.var + ro) : zero;
    // the slots are only needed for the update at the end, but asked for here: behind the slot walk their round trip
    // would be the last link of every wave's chain
    float4 m = zero, v = zero;
    if (g == 0 && act) {
        m = ld4(t.m + ro);
        v = ld4(t.v + ro);
    }
    float4 acc = zero;
    int touched = 0;
    const auto walk1 = [&](int2 rs) __attribute__((always_inline)) { // a per-epoch index (one chunk): most rows hold a slot or two
        touched += rs.y;
        for (int k = rs.x + g; k < rs.x + rs.y; k += G) {
            if (act) {
                const float4 cc = ld4(contrib + (int64_t)order[k] * d + 4 * l);
                acc.x = acc.x + (cc.x + o.reg_occ * var.x);
                acc.y = acc.y + (cc.y + o.reg_occ * var.y);
                acc.z = acc.z + (cc.z + o.reg_occ * var.z);
                acc.w = acc.w + (cc.w + o.reg_occ * var.w);
            }
        }
    };
    // A per-step index (several chunks: the item tables of CML / LRML / SBPR, whose popular rows hold tens of slots): WALK slots of a
    // group at a time, their ids, then their contribution rows, in flight together -- such a row's walk is the longest chain of
    // the launch (stamps: 11.6 us for the item rows with more than 32 slots one slot at a time, 10.2 three at a time).
    constexpr int WALK = 3;
    const auto walk = [&](int2 rs) __attribute__((always_inline)) {
        touched += rs.y;
        const int end = rs.x + rs.y;
        for (int k = rs.x + g; k < end; k += WALK * G) {
            if (act) {
                int sl[WALK];
                float4 cc[WALK];
#pragma unroll
                for (int e = 0; e < WALK; ++e)
                    sl[e] = k + e * G < end ? order[k + e * G] : 0;
#pragma unroll
                for (int e = 0; e < WALK; ++e)
                    cc[e] = k + e * G < end ? ld4(contrib + (int64_t)sl[e] * d + 4 * l) : zero;
#pragma unroll
                for (int e = 0; e < WALK; ++e)
                    if (k + e * G < end) {
                        acc.x = acc.x + (cc[e].x + o.reg_occ * var.x);
                        acc.y = acc.y + (cc[e].y + o.reg_occ * var.y);
                        acc.z = acc.z + (cc[e].z + o.reg_occ * var.z);
                        acc.w = acc.w + (cc[e].w + o.reg_occ * var.w);
                    }
            }
        }
    };
    if (!CHUNKED || chunks == 1) // (CHUNKED: the launch was built for a set with a per-step index; the other launches do not carry the second walk)
        walk1(rs0);
    else
        walk(rs0);
    // the other chunks, in ascending slot order; the segments of SEGS chunks are asked for together (one round trip, not one
    // per chunk: a row with its slots in nine chunks waited nine times before its last contribution)
    for (int c0 = 1; c0 < chunks; c0 += SEGS) {
        int2 rsv[SEGS];
#pragma unroll
        for (int c = 0; c < SEGS; ++c)
            rsv[c] = c0 + c < chunks ? rowseg[(int64_t)(c0 + c) * t.rows + r] : make_int2(0, 0);
#pragma unroll
        for (int c = 0; c < SEGS; ++c) {
            if (CHUNKED)
                walk(rsv[c]);
            else
                walk1(rsv[c]);
        }
    }
    ROWS_MARK(5, __float_as_int(acc.x) ^ touched); // every slot walked
    ROWS_NOTE(2, touched);
    if (touched == 0 && untouched_stays(o))
        return;
    ROWS_MARK(6, __float_as_int(var.x)); // the row itself has arrived (asked for before the walk)
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
    if (o.row_norm) {
        float sq = (var.x * var.x + var.y * var.y) + (var.z * var.z + var.w * var.w);
#pragma unroll
        for (int off = LPR >> 1; off > 0; off >>= 1)
            sq += __shfl_xor(sq, off, 64);
        if (row_out && lane == 0)
            row_out[r] = sq;
    }
    if (g != 0 || !act)
        return;
    float4 nv = var;
    // component by component in registers (indexing the float4s through pointers put them in scratch)
    const auto upd = [&](float g_, float d_, float var_, float &m_, float &v_, float &n_) {
        const bool dense = o.kind == ROWS_DENSE;
        n_ = opt_elem(o, dense, dense ? (g_ + d_) + o.dense_reg * var_ : g_, var_, m_, v_);
    };
    upd(acc.x, dg.x, var.x, m.x, v.x, nv.x);
    upd(acc.y, dg.y, var.y, m.y, v.y, nv.y);
    upd(acc.z, dg.z, var.z, m.z, v.z, nv.z);
    upd(acc.w, dg.w, var.w, m.w, v.w, nv.w);
    ROWS_MARK(7, __float_as_int(nv.x) ^ __float_as_int(m.x)); // m, v arrived, the optimizer done
    st4_stream(t.var + ro, nv);
    st4_stream(t.m + ro, m);
    st4_stream(t.v + ro, v);
}

// Two tables that share their row ids AND their slots (NeuMF's P_gmf / P_mlp of a user, Q_gmf / Q_mlp of an item: every
// instance gathers the same row of both) updated by ONE wave: the row's segments and slot ids are walked once, the two
// contribution rows of a slot travel together -- half the waves, twice the bytes in flight per wave.  TF1 sparse Adam only.
template <int LPR>
__device__ __forceinline__ void rows_apply_row_pair(const crec_rows_table &t1, const crec_rows_table &t2, const int2 *__restrict__ rowseg,
                                                    const int32_t *__restrict__ order, const float *__restrict__ contrib1,
                                                    const float *__restrict__ contrib2, const crec_rows_opt &o, float reg2, int64_t r,
                                                    int chunks)
{
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    if (r >= t1.rows)
        return;
    const int d = t1.d;
    const bool act = 4 * l < d;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t ro = r * d + 4 * l;
    const int2 rs0 = rowseg[r];
    const float4 var1 = act ? ld4(t1.var + ro) : zero, var2 = act ? ld4(t2.var + ro) : zero;
    float4 m1 = zero, v1 = zero, m2 = zero, v2 = zero;
    if (g == 0 && act) {
        m1 = ld4(t1.m + ro), v1 = ld4(t1.v + ro);
        m2 = ld4(t2.m + ro), v2 = ld4(t2.v + ro);
    }
    float4 a1 = zero, a2 = zero;
    for (int ch = 0; ch < chunks; ++ch) { // chunk by chunk = ascending slot order
        int2 rs = rs0;
        if (ch != 0)
            rs = rowseg[(int64_t)ch * t1.rows + r];
        for (int k = rs.x + g; k < rs.x + rs.y; k += G) {
            if (act) {
                const int64_t so = (int64_t)order[k] * d + 4 * l;
                const float4 c1 = ld4(contrib1 + so), c2 = ld4(contrib2 + so);
                a1.x = a1.x + (c1.x + o.reg_occ * var1.x), a1.y = a1.y + (c1.y + o.reg_occ * var1.y);
                a1.z = a1.z + (c1.z + o.reg_occ * var1.z), a1.w = a1.w + (c1.w + o.reg_occ * var1.w);
                a2.x = a2.x + (c2.x + reg2 * var2.x), a2.y = a2.y + (c2.y + reg2 * var2.y);
                a2.z = a2.z + (c2.z + reg2 * var2.z), a2.w = a2.w + (c2.w + reg2 * var2.w);
            }
        }
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) { // groups hold disjoint slots: add them, lower group first
        a1.x = a1.x + __shfl_down(a1.x, off, 64), a1.y = a1.y + __shfl_down(a1.y, off, 64);
        a1.z = a1.z + __shfl_down(a1.z, off, 64), a1.w = a1.w + __shfl_down(a1.w, off, 64);
        a2.x = a2.x + __shfl_down(a2.x, off, 64), a2.y = a2.y + __shfl_down(a2.y, off, 64);
        a2.z = a2.z + __shfl_down(a2.z, off, 64), a2.w = a2.w + __shfl_down(a2.w, off, 64);
    }
    if (g != 0 || !act)
        return;
    const float omb1 = 1.0f - o.beta1, omb2 = 1.0f - o.beta2;
    const auto upd = [&](float g_, float var_, float &m_, float &v_) {
        m_ = m_ * o.beta1 + g_ * omb1;
        v_ = v_ * o.beta2 + (g_ * g_) * omb2;
        return var_ - (o.lr_t * m_) / (sqrtf(v_) + o.eps);
    };
    float4 n1, n2;
    n1.x = upd(a1.x, var1.x, m1.x, v1.x), n1.y = upd(a1.y, var1.y, m1.y, v1.y), n1.z = upd(a1.z, var1.z, m1.z, v1.z), n1.w = upd(a1.w, var1.w, m1.w, v1.w);
    n2.x = upd(a2.x, var2.x, m2.x, v2.x), n2.y = upd(a2.y, var2.y, m2.y, v2.y), n2.z = upd(a2.z, var2.z, m2.z, v2.z), n2.w = upd(a2.w, var2.w, m2.w, v2.w);
    st4_stream(t1.var + ro, n1), st4_stream(t1.m + ro, m1), st4_stream(t1.v + ro, v1);
    st4_stream(t2.var + ro, n2), st4_stream(t2.m + ro, m2), st4_stream(t2.v + ro, v2);
}

// The same update with the wave's G = 64/LPR lane groups SIDE BY SIDE, one table row each (rows r0 .. r0+G-1): every
// lane does Adam arithmetic and a wave moves G rows per round trip instead of one -- the sweep over a table is bound by
// rounds of waves times a wave's chain of round trips, not by bytes.  A group adds its row's slots in ascending slot
// order, four contribution rows in flight.  A wave that holds a row with more than SIDE_MAX slots (popular items) takes
// its rows one after the other in the whole-wave form above instead, where the groups share a row's slots.
constexpr int SIDE_MAX = 12;
// measured (tools/bench_models.py): side by side wins at d = 32 (FM C4: 19.2 -> 16.7 us/step, eight rows per wave) and
// loses 3-10 % at d = 64 / 128 (NeuMF C3, GMF, CML: four / two rows per wave), where a wave per row stays
constexpr int rows_per_wave(int lpr) { return lpr <= 8 ? 64 / lpr : 1; }
template <int LPR>
__device__ __forceinline__ void rows_apply_side(const crec_rows_table &t, const int2 *__restrict__ rowseg,
                                                const int32_t *__restrict__ order, const float *__restrict__ contrib,
                                                const float *__restrict__ dense_grad, float *__restrict__ row_out,
                                                const crec_rows_opt &o, int64_t r0, int chunks)
{
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const int64_t r = r0 + g;
    const bool valid = r < t.rows;
    int cnt = 0;
    if (valid)
        for (int ch = 0; ch < chunks; ++ch)
            cnt += rowseg[(int64_t)ch * t.rows + r].y;
    int wmax = cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int other = __shfl_xor(wmax, off, 64);
        wmax = other > wmax ? other : wmax;
    }
    if (G == 1 || wmax > SIDE_MAX) { // wave-uniform
        for (int k = 0; k < G; ++k)
            rows_apply_row<LPR>(t, rowseg, order, contrib, dense_grad, row_out, o, r0 + k, chunks);
        return;
    }
    const int d = t.d;
    const bool act = valid && 4 * l < d;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cnt == 0 && untouched_stays(o))
        return;
    const int64_t ro = r * d + 4 * l;
    const float4 var = act ? ld4(t.var + ro) : zero;
    float4 m = act ? ld4(t.m + ro) : zero, v = act ? ld4(t.v + ro) : zero;
    float4 acc = zero;
    if (act) {
        for (int ch = 0; ch < chunks; ++ch) { // chunk by chunk = ascending slot order
            const int2 rs = rowseg[(int64_t)ch * t.rows + r];
            for (int k = rs.x; k < rs.x + rs.y; k += 4) {
                float4 c[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    c[e] = k + e < rs.x + rs.y ? ld4(contrib + (int64_t)order[k + e] * d + 4 * l) : zero;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (k + e < rs.x + rs.y) {
                        acc.x = acc.x + (c[e].x + o.reg_occ * var.x);
                        acc.y = acc.y + (c[e].y + o.reg_occ * var.y);
                        acc.z = acc.z + (c[e].z + o.reg_occ * var.z);
                        acc.w = acc.w + (c[e].w + o.reg_occ * var.w);
                    }
            }
        }
    }
    float4 dg = zero;
    if (o.cov_coef != 0.f) { // CML covariance term, see rows_apply_row
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
        if (row_out && valid && l == 0)
            row_out[r] = sr * sr - sq;
    } else if (dense_grad && act) {
        dg = ld4(dense_grad + ro);
    }
    if (o.row_norm) {
        float sq = (var.x * var.x + var.y * var.y) + (var.z * var.z + var.w * var.w);
#pragma unroll
        for (int off = LPR >> 1; off > 0; off >>= 1)
            sq += __shfl_xor(sq, off, 64);
        if (row_out && valid && l == 0)
            row_out[r] = sq;
    }
    if (!act)
        return;
    float4 nv = var;
    const auto upd = [&](float g_, float d_, float var_, float &m_, float &v_, float &n_) {
        const bool dense = o.kind == ROWS_DENSE;
        n_ = opt_elem(o, dense, dense ? (g_ + d_) + o.dense_reg * var_ : g_, var_, m_, v_);
    };
    upd(acc.x, dg.x, var.x, m.x, v.x, nv.x);
    upd(acc.y, dg.y, var.y, m.y, v.y, nv.y);
    upd(acc.z, dg.z, var.z, m.z, v.z, nv.z);
    upd(acc.w, dg.w, var.w, m.w, v.w, nv.w);
    st4_stream(t.var + ro, nv);
    st4_stream(t.m + ro, m);
    st4_stream(t.v + ro, v);
}

template <int LPR>
__global__ __launch_bounds__(256) void rows_apply(crec_rows_table t, const int2 *__restrict__ rowseg,
                                                  const int32_t *__restrict__ order, const float *__restrict__ contrib,
                                                  const float *__restrict__ dense_grad, float *__restrict__ row_out,
                                                  crec_rows_opt o, int chunks)
{
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (rows_per_wave(LPR) > 1)
        rows_apply_side<LPR>(t, rowseg, order, contrib, dense_grad, row_out, o, wave * (64 / LPR), chunks);
    else
        rows_apply_row<LPR>(t, rowseg, order, contrib, dense_grad, row_out, o, wave, chunks);
}

// dense ApplyAdam of a small parameter vector (h, W_k, b_k, w0) from partial sums:
//   g[e] = sum_p partial[p*stride + e], n elements.  A workgroup of EL * SAD_SL threads owns EL elements; its
//   SAD_SL slices each add a contiguous run of the partials in ascending p (8 independent loads in flight at a
//   time), and the slice sums are combined in slice order -- one fixed summation order whatever the launch shape.
constexpr int SAD_SL = 16;
template <int EL> __device__ __forceinline__ void small_adam_block(const crec_small_job &j, int blk, float *red)
{
    const int el = threadIdx.x % EL, sl = threadIdx.x / EL;
    const int e = blk * EL + el;
    const int per = (j.n_partials + SAD_SL - 1) / SAD_SL;
    const int p0 = sl * per, p1 = (p0 + per) < j.n_partials ? (p0 + per) : j.n_partials;
    float g = 0.f;
    if (e < j.n) {
        int p = p0;
        for (; p + 8 <= p1; p += 8) {
            float t[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                t[k] = j.partial[(int64_t)(p + k) * j.stride + e];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                g = g + t[k];
        }
        for (; p < p1; ++p)
            g = g + j.partial[(int64_t)p * j.stride + e];
    }
    red[sl * EL + el] = g;
    __syncthreads();
    if (sl != 0 || e >= j.n)
        return;
    g = red[el];
#pragma unroll
    for (int s = 1; s < SAD_SL; ++s)
        g = g + red[s * EL + el];
    float mm = j.m[e], vv = j.v[e];
    const float nv = opt_elem(j.o, true, g, j.var[e], mm, vv);
    j.m[e] = mm;
    j.v[e] = vv;
    j.var[e] = nv;
}

__global__ __launch_bounds__(64 * SAD_SL) void small_adam_dense(crec_small_job j)
{
    __shared__ float red[SAD_SL * 64];
    small_adam_block<64>(j, blockIdx.x, red);
}

// a one-column table: thread per row
__device__ __forceinline__ void scalar_rows_block(const crec_scalar_job &j, int blk)
{
    const int64_t r = (int64_t)blk * 256 + threadIdx.x;
    if (r >= j.rows)
        return;
    float g = 0.f;
    int touched = 0;
    const int chunks = j.chunks > 1 ? j.chunks : 1;
    for (int c0 = 0; c0 < chunks; c0 += SEGS) { // chunk by chunk = ascending slot order; SEGS segments asked for together
        int2 rsv[SEGS];
#pragma unroll
        for (int c = 0; c < SEGS; ++c)
            rsv[c] = c0 + c < chunks ? j.rowseg[(int64_t)(c0 + c) * j.rows + r] : make_int2(0, 0);
#pragma unroll
        for (int c = 0; c < SEGS; ++c)
            touched += rsv[c].y;
#pragma unroll
        for (int c = 0; c < SEGS; ++c)
            for (int k = rsv[c].x; k < rsv[c].x + rsv[c].y; k += 4) { // four slots' ids, then their four terms, in flight
                int id[4];
                float x[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    id[e] = k + e < rsv[c].x + rsv[c].y ? j.order[k + e] : -1;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    x[e] = id[e] >= 0 ? j.contrib[id[e]] : 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (id[e] >= 0)
                        g = g + x[e];
            }
    }
    if (touched == 0 && j.o.kind == ROWS_SPARSE_LAZY) // lazy Adam: an element no slot names stays, slots included
        return;
    const float w = j.var[r];
    if (j.sq_out)
        j.sq_out[r] = w * w;
    g = g + j.o.dense_reg * w;
    float mm = j.m[r], vv = j.v[r];
    const float nv = opt_elem(j.o, true, g, w, mm, vv);
    j.m[r] = mm;
    j.v[r] = vv;
    j.var[r] = nv;
}

// every parameter update of a step in ONE launch: up to CREC_APPLY_MAX_TABLES embedding tables (blocks
// [block_end[k-1], block_end[k]) sweep table k, four rows per block), then an optional one-column table (256 rows
// per block); the small dense vector leads the grid
// (16 elements per block; same slices, hence the same sums, as small_adam_dense)
template <int LPR, bool CHUNKED> __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void apply_all(crec_apply_set a)
{
    __shared__ float red[SAD_SL * 16];
#ifdef CREC_ROWS_STAMPS
    rows_mark(0, 0, false);
    struct End {
        __device__ ~End()
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            rows_mark(1, 0, false);
        }
    } end_;
#endif
    // the small vector's blocks come FIRST: theirs is the longest chain of the launch (two rounds of strided partial
    // loads, a reduction, then the update), which then runs beside the table sweep instead of after it
    if ((int)blockIdx.x < a.small_blocks) {
        small_adam_block<16>(a.small, blockIdx.x, red);
        return;
    }
    const int bi = (int)blockIdx.x - a.small_blocks;
    int k = 0, first = 0;
#pragma unroll
    for (int q = 0; q < CREC_APPLY_MAX_TABLES; ++q) // unrolled: a loop indexing block_end[] sent the array to scratch
        if (q < a.n_tables && bi >= a.block_end[q]) {
            first = a.block_end[q];
            k = q + 1;
        }
    if (k < a.n_tables) {
        const crec_rows_job &j = a.tables[k];
        const int64_t wave = (int64_t)(bi - first) * 4 + (threadIdx.x >> 6);
        if (j.t2.var)
            rows_apply_row_pair<LPR>(j.t, j.t2, j.rowseg, j.order, j.contrib, j.contrib2, j.o, j.reg_occ2, wave, j.chunks > 1 ? j.chunks : 1);
        else if (rows_per_wave(LPR) > 1)
            rows_apply_side<LPR>(j.t, j.rowseg, j.order, j.contrib, j.dense_grad, j.row_out, j.o, wave * (64 / LPR),
                                 j.chunks > 1 ? j.chunks : 1);
        else
            rows_apply_row<LPR, CHUNKED>(j.t, j.rowseg, j.order, j.contrib, j.dense_grad, j.row_out, j.o, wave, j.chunks > 1 ? j.chunks : 1);
    } else {
        scalar_rows_block(a.scalar, bi - first);
    }
}

} // namespace

size_t crec_rows_index_bytes(int64_t n_batches, int32_t per_batch, int64_t rows, int32_t chunks)
{
    return sizeof(int32_t) * (size_t)n_batches * per_batch + sizeof(int2) * (size_t)n_batches * chunks * rows + 512;
}

int crec_rows_index(crec_ctx *ctx, const int32_t *idx, int64_t n_total, int32_t per_batch, int64_t rows, void *ws,
                    crec_rows_index_view *out, hipStream_t st, int32_t chunks)
{
    CREC_REQUIRE(ctx, chunks >= 1 && chunks <= 64, "rows index: 1..64 chunks");
    const int clen = (per_batch + chunks - 1) / chunks;
    CREC_REQUIRE(ctx, per_batch >= 1 && clen <= 16384, "rows index: at most 16384 slots per chunk, got %d", clen);
    const int64_t nb = crec_ceil_div(n_total, per_batch);
    out->order = (int32_t *)ws;
    out->rowseg = (int2 *)((char *)ws + ((sizeof(int32_t) * (size_t)nb * per_batch + 255) / 256) * 256);
    out->per_batch = per_batch;
    out->rows = rows;
    out->chunks = chunks;
    const int cap = (clen + 63) / 64 * 64;
    const size_t lds = rows_index_lds_bytes(cap);
    int key_bits = 1;
    while (key_bits < 32 && ((int64_t)1 << key_bits) < rows)
        ++key_bits;
    CREC_HIP(ctx, hipFuncSetAttribute((const void *)rows_index_batches, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    crec_prof_begin(ctx, CREC_K_ROWS_INDEX, st);
    CREC_KLAUNCH(ctx, rows_index_batches, dim3((unsigned)(nb * chunks)), dim3(IDX_THREADS), lds, st, idx, n_total, per_batch,
                       chunks, rows, key_bits, cap, out->order, out->rowseg);
    crec_prof_end(ctx, st);
    return CREC_OK;
}

int crec_rows_apply(crec_ctx *ctx, const crec_rows_table &t, const crec_rows_index_view &ix, int64_t batch,
                    const float *contrib, const float *dense_grad, const crec_rows_opt &o, hipStream_t st, float *row_out)
{
    int lpr = 1;
    while (lpr * 4 < t.d)
        lpr <<= 1;
    const int2 *rs = ix.rowseg + batch * ix.chunks * ix.rows;
    const int32_t *ord = ix.order + batch * ix.per_batch;
    const dim3 grid((unsigned)crec_ceil_div(t.rows, 4 * rows_per_wave(lpr))), block(256);
    crec_prof_begin(ctx, CREC_K_ROWS_APPLY, st);
#define CREC_RA(L)                                                                                   \
    case L:                                                                                          \
        CREC_KLAUNCH(ctx, (rows_apply<L>), grid, block, 0, st, t, rs, ord, contrib, dense_grad, row_out, o, (int)ix.chunks); \
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
    CREC_KLAUNCH(ctx, small_adam_dense, dim3((unsigned)crec_ceil_div(n, 64)), dim3(64 * SAD_SL), 0, st,
                       crec_small_job{var, m, v, partial, n_partials, stride, n, o});
    crec_prof_end(ctx, st);
    return CREC_OK;
}

int crec_apply_all(crec_ctx *ctx, crec_apply_set &a, hipStream_t st)
{
    CREC_REQUIRE(ctx, a.n_tables >= 0 && a.n_tables <= CREC_APPLY_MAX_TABLES, "apply set: too many tables");
#ifdef CREC_ROWS_STAMPS // timing diagnosis only, debug builds (tools/rows_stamps.sh, tools/time_cml.py): CREC_APPLY_SKIP=k leaves table k of
                        // every update launch out.  Never in the shipped library: a stray environment variable must not change training
    static const int skip = getenv("CREC_APPLY_SKIP") ? atoi(getenv("CREC_APPLY_SKIP")) : -1;
    if (skip >= 0 && skip < a.n_tables)
        a.tables[skip].t.rows = 0;
#endif
    int lpr = 0, blocks = 0;
    bool chunked = false; // some table of the set comes with a per-step index in several chunks (rows_apply_row's second walk)
    for (int k = 0; k < a.n_tables; ++k) {
        chunked = chunked || (a.tables[k].chunks > 1 && !a.tables[k].t2.var && a.tables[k].t.rows > 0);
        int l = 1;
        while (l * 4 < a.tables[k].t.d)
            l <<= 1;
        CREC_REQUIRE(ctx, lpr == 0 || l == lpr, "apply set: tables of one launch must share the row width class");
        lpr = l;
        CREC_REQUIRE(ctx, !a.tables[k].t2.var || (a.tables[k].t2.rows == a.tables[k].t.rows && a.tables[k].t2.d == a.tables[k].t.d &&
                                                   a.tables[k].o.kind == ROWS_SPARSE_TF1 && a.tables[k].o.opt == 0 && !a.tables[k].dense_grad && !a.tables[k].row_out),
                     "apply set: a paired table shares rows, width and slots with its partner (TF1 sparse Adam only)");
        blocks += (int)crec_ceil_div(a.tables[k].t.rows, 4 * (a.tables[k].t2.var ? 1 : rows_per_wave(l)));
        a.block_end[k] = blocks;
    }
    if (a.scalar.rows > 0)
        blocks += (int)crec_ceil_div(a.scalar.rows, 256);
    a.scalar_end = blocks;
    a.small_blocks = a.small.n > 0 ? (int)crec_ceil_div(a.small.n, 16) : 0;
    blocks += a.small_blocks;
    if (blocks == 0)
        return CREC_OK;
    if (lpr == 0)
        lpr = 1;
    crec_prof_begin(ctx, CREC_K_ROWS_APPLY, st);
#define CREC_AA(L)                                                                  \
    case L:                                                                         \
        if (chunked)                                                                \
            CREC_KLAUNCH(ctx, (apply_all<L, true>), dim3((unsigned)blocks), dim3(256), 0, st, a);  \
        else                                                                        \
            CREC_KLAUNCH(ctx, (apply_all<L, false>), dim3((unsigned)blocks), dim3(256), 0, st, a); \
        break;
    switch (lpr) {
        CREC_AA(1)
        CREC_AA(2)
        CREC_AA(4)
        CREC_AA(8)
        CREC_AA(16)
        CREC_AA(32)
        CREC_AA(64)
    }
#undef CREC_AA
    crec_prof_end(ctx, st);
    return CREC_OK;
}
