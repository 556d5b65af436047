// Full-catalogue evaluation for gfx950: scores = P[users] . Q^T on the fp32 matrix cores with
// the seen-item filter and the top-K selection fused behind the accumulators, so the
// [users, items] score matrix never exists in memory.
//
// Replaces sess.run(pre_scores) (model/ranking/BPR.py:51) + np.argsort + the seen-item loop of
// RankingRecommender.test_model_rs (model/RankingRecommender.py:221-240) and cal_ranking_metrics
// (utils/metrics.py:9-19); see include/crec.h.
//
// eval_sweep<DMAX, TI>: a workgroup = 4 waves; wave w owns 32 users (the rows of a
//   v_mfma_f32_32x32x2_f32 tile) and keeps their embeddings in DMAX/2 registers per lane for
//   the whole sweep; the 4 waves share item tiles of TI rows staged global -> registers -> LDS
//   (rows padded by 16 B: conflict-free ds_read_b128), double buffered.  MFMA step s multiplies
//   k = s (lanes 0-31) and k = DMAX/2 + s (lanes 32-63), so a lane reads its B operands as
//   contiguous float4s; the CPU oracle's order == 1 folds products in that same order, which
//   makes the fp32 scores bit-identical.
//   Behind each 32x32 accumulator tile: score > the user's current K-th best (16 compares per
//   lane, the thresholds live in registers) -> rare candidates are appended to a per-user
//   64-slot buffer in LDS; a full buffer is pruned by the wave: binary search of each new
//   candidate in the user's sorted seen list, rank by all-pairs compare, keep the K best,
//   raise the threshold.  Keys are (ordered score bits, ~item id) so ties go to the lower id.
//   grid.y splits the catalogue; each split writes K keys per user.
// eval_merge: one wave per user ranks the splits' keys and writes ids / scores.
// rank_metrics: HR / MRR / NDCG per (K, user) in float64.
#include "crec_internal.h"

#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

// Candidate slots per user: after a prune at most K stay, and one block can add 32 (one per item lane), so
// CAP >= K + 32.  Two shapes: 4 waves x CAP 64 (K <= 32), or 8 waves x CAP 56 (K <= 24) -- two waves per SIMD, so that
// one wave's epilogue, LDS reads and barrier waits fill the other's MFMA gaps -- when its buffers fit the 160 KB of LDS.
constexpr int K_MAX = 32;       // candidates the sweep keeps per user and pass
constexpr int K_LIST_MAX = 128; // the longest list crec_eval_topk ranks (four passes) and crec_rank_metrics reads
constexpr int cap_of(int NW) { return NW == 8 ? 56 : 64; }
constexpr int MAX_SPLITS = 16;

__device__ inline uint32_t ordered_bits(float f)
{
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ inline float from_ordered(uint32_t o)
{
    uint32_t u = o ^ ((o >> 31) ? 0x80000000u : 0xffffffffu);
    return __uint_as_float(u);
}
__device__ inline u64 make_key(float score, int item) { return ((u64)ordered_bits(score) << 32) | (0xffffffffu - (uint32_t)item); }
__device__ inline int key_item(u64 k) { return (int)(0xffffffffu - (uint32_t)k); }
__device__ inline float key_score(u64 k) { return from_ordered((uint32_t)(k >> 32)); }

__device__ inline void lds_fence()
{
    // LDS operations of one wave execute and return in issue order, so the cross-lane traffic below needs no
    // hardware wait of its own; this only stops the compiler from caching or reordering LDS accesses across it.
    // (A wavefront-scope fence would also drain vmcnt -- i.e. wait for the next tile's global loads in flight.)
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

struct WaveState {
    u64 *cand;   // [32][cap]
    int *cnt;    // [32] entries in the buffer
    int *clean;  // [32] leading entries already checked against the seen list
    float *thr;  // [32] score a candidate must beat
    int *uid;    // [32] global user id, -1 for padding rows
    int cap;     // candidate slots per user
    const uint32_t *bits; // optional [rows][words] membership bitmap of the users' lists (small catalogues), else nullptr
    int64_t row0;         // bitmap row of this wave's first user
    int words;
    int keep_listed; // 0: drop items found in the user's list (seen filter); 1: keep only those (candidates)
    const int32_t *excl;  // K > 32, list mode: [rows][excl_stride] the ids the earlier passes have ranked (first n_excl per row)
    int n_excl, excl_stride;
};

// Keep the K best unseen entries of user row `ul` (n buffered), sorted best first; returns how many stay.  Whole wave.
__device__ inline int prune(const WaveState &ws, int ul, int K, int n, const int64_t *__restrict__ seen_ptr,
                            const int32_t *__restrict__ seen_items)
{
    const int lane = threadIdx.x & 63;
    u64 *buf = ws.cand + ul * ws.cap;
    u64 mine = lane < n ? buf[lane] : 0;
    if (!ws.bits) { // with bitmaps the candidates were filtered before they got here
        const int clean = ws.clean[ul], uid = ws.uid[ul];
        if (lane >= clean && lane < n) {
            const int item = key_item(mine);
            int64_t lo = seen_ptr[uid], hi = seen_ptr[uid + 1];
            const int64_t end = hi;
            while (lo < hi) {
                int64_t mid = (lo + hi) >> 1;
                if (seen_items[mid] < item)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            // exclude mode: listed = seen in training, never recommended (RankingRecommender.py:236);
            // keep mode: listed = the user's candidate items (leave-one-out / sampled-negative evaluation, :250-299)
            const bool listed = lo < end && seen_items[lo] == item;
            if (listed != (ws.keep_listed != 0))
                mine = 0;
            for (int q = 0; q < ws.n_excl; ++q) // the ranks above this pass's (a list longer than 32 is ranked 32 at a time)
                if (ws.excl[(ws.row0 + ul) * ws.excl_stride + q] == item)
                    mine = 0;
        }
        if (lane < ws.cap) // a user's buffer may be shorter than the wave
            buf[lane] = mine;
    }
    lds_fence();
    int rank = 0;
    // all-pairs rank against the (<= 64) buffered keys: broadcast LDS reads, kept in flight eight at a time -- one at
    // a time this loop was most of a prune (56 x an LDS round trip)
    int t = 0;
    for (; t + 8 <= n; t += 8) {
        u64 k[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            k[q] = buf[t + q];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            rank += k[q] > mine;
    }
    for (; t < n; ++t)
        rank += buf[t] > mine;
    lds_fence();
    const bool valid = mine != 0;
    const int n_valid = __popcll(__ballot(valid));
    if (valid && rank < K)
        buf[rank] = mine;
    const int kept = n_valid < K ? n_valid : K;
    if (n_valid >= K) {
        if (valid && rank == K - 1)
            ws.thr[ul] = key_score(mine);
    }
    if (lane == 0 && !ws.bits)
        ws.clean[ul] = kept;
    lds_fence();
    return kept;
}

template <int DMAX, int TI, int NW>
__global__ __launch_bounds__(NW * 64) void eval_sweep(const float *__restrict__ P, const float *__restrict__ Q, int64_t I,
                                                  int d, const int32_t *__restrict__ users, int Bt,
                                                  const int64_t *__restrict__ seen_ptr,
                                                  const int32_t *__restrict__ seen_items, int K, float sign,
                                                  int64_t items_per_split, u64 *__restrict__ part /*[Bt][S][K]*/,
                                                  const uint32_t *__restrict__ bits, int words, const int32_t *__restrict__ excl, int n_excl,
                                                  int excl_stride)
{
    constexpr int HS = DMAX / 2;          // MFMA steps
    constexpr int LDQ = DMAX + 4;         // padded LDS row (floats)
    constexpr int CPR = DMAX / 4;         // float4 chunks per item row
    constexpr int NT = NW * 64, CAP = cap_of(NW), USERS_PER_WG = NW * 32;
    constexpr int NCH = TI * CPR / NT;    // chunks staged per thread
    constexpr int NBLK = TI / 32;
    static_assert(NCH >= 1, "tile too small");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *qt = (float *)smem;                                  // [2][TI*LDQ]
    u64 *cand_all = (u64 *)(qt + 2 * TI * LDQ);                 // [NW][32][CAP]
    int *cnt_all = (int *)(cand_all + NW * 32 * CAP);           // [NW][32]
    int *clean_all = cnt_all + NW * 32;
    float *thr_all = (float *)(clean_all + NW * 32);
    int *uid_all = (int *)(thr_all + NW * 32);

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int row = lane & 31, h = lane >> 5;
    // the host packs the filter mode above K: K | keep_listed << 16
    WaveState ws{cand_all + wv * 32 * CAP, cnt_all + wv * 32, clean_all + wv * 32, thr_all + wv * 32, uid_all + wv * 32, CAP, bits,
                 (int64_t)blockIdx.x * USERS_PER_WG + wv * 32, words, K >> 16, excl, n_excl, excl_stride};
    K &= 0xffff;

    const int S = gridDim.y, sp = blockIdx.y;
    const int64_t i0 = (int64_t)sp * items_per_split;
    const int64_t i1 = (i0 + items_per_split) < I ? (i0 + items_per_split) : I;

    // ---- per-wave state + the A operand (this wave's 32 users) --------------------------
    const int urow = blockIdx.x * USERS_PER_WG + wv * 32 + row;
    const int uid = urow < Bt ? users[urow] : -1;
    if (h == 0) {
        ws.cnt[row] = 0;
        ws.clean[row] = 0;
        ws.thr[row] = uid >= 0 ? -INFINITY : INFINITY; // padding rows never produce candidates
        ws.uid[row] = uid;
    }
    float a[HS];
#pragma unroll
    for (int q = 0; q < HS / 4; ++q) {
        const int k = h * HS + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (uid >= 0 && k < d)
            v = *reinterpret_cast<const float4 *>(P + (int64_t)uid * d + k);
        a[4 * q + 0] = sign * v.x;
        a[4 * q + 1] = sign * v.y;
        a[4 * q + 2] = sign * v.z;
        a[4 * q + 3] = sign * v.w;
    }
    // zero both tile buffers once: the k >= d padding columns are never written again
    for (int k = tid; k < 2 * TI * LDQ; k += NT)
        qt[k] = 0.f;
    __syncthreads();

    // ---- staging helpers --------------------------------------------------------------------
    float4 stage[NCH];
    auto stage_load = [&](int64_t tile_base) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = tid + c * NT, r = ch / CPR, k = (ch % CPR) * 4;
            const int64_t item = tile_base + r;
            stage[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (item < i1 && k < d)
                stage[c] = *reinterpret_cast<const float4 *>(Q + item * d + k);
        }
    };
    auto stage_store = [&](int bufi) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int ch = tid + c * NT, r = ch / CPR, k = (ch % CPR) * 4;
            if (k < d)
                *reinterpret_cast<float4 *>(qt + bufi * TI * LDQ + r * LDQ + k) = stage[c];
        }
    };

    float thr_r[16];
    auto reload_thr = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            thr_r[r] = ws.thr[(r & 3) + 8 * (r >> 2) + 4 * h];
    };
    lds_fence();
    reload_thr();

    const int64_t n_tiles = i1 > i0 ? (i1 - i0 + TI - 1) / TI : 0;
    if (n_tiles > 0) {
        stage_load(i0);
        stage_store(0);
    }
    __syncthreads();

    // Software pipeline over the stream of 32x32 blocks: while the matrix core works through block n (a chain of
    // DMAX/2 dependent MFMAs), the wave's vector ALU compares block n-1's accumulators with the thresholds -- the
    // compares sit between the MFMAs in program order, so they issue in the MFMAs' shadow instead of after them.
    // Two accumulators alternate along the stream (the tile body is instantiated for both parities of its first
    // block).  Candidates are rare: one ballot per block decides whether the append / prune path runs at all.
    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        accA[r] = accB[r] = 0.f;

    // append block candidates (bits of `hit`) of block `item0 + row`; prunes full buffers
    int cnt_reg = 0; // lane ul < 32: entries in user ul's candidate buffer (kept in a register: no LDS round trip per append)
    auto take_candidates = [&](const f32x16 acc, uint32_t hit, int64_t item) __attribute__((always_inline)) {
        bool pruned = false;
        if (ws.bits) {
            // small catalogues: drop listed (or keep only listed) items HERE, with one coalesced load of the 32 users'
            // bitmap words for this 32-item block, so that a prune is pure LDS work.  Lane L < 32 holds user L's word.
            const int64_t wi = (item - row) >> 5; // blocks are 32-aligned: one word per (user, block)
            const uint32_t wmine = (lane < 32 && ws.uid[lane] >= 0) ? ws.bits[(ws.row0 + lane) * ws.words + wi] : 0u;
            const uint32_t keep = ws.keep_listed ? 1u : 0u;
            uint32_t ok = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t wr = (uint32_t)__shfl((int)wmine, (r & 3) + 8 * (r >> 2) + 4 * h, 64);
                ok |= (((wr >> row) & 1u) == keep ? 1u : 0u) << r;
            }
            hit &= ok;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool is_cand = (hit >> r) & 1u;
            const u64 bal = __ballot(is_cand);
            if (bal) {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const uint32_t m = hh ? (uint32_t)(bal >> 32) : (uint32_t)bal;
                    if (m) {
                        const int ul = (r & 3) + 8 * (r >> 2) + 4 * hh;
                        const int c = __popc(m);
                        int base = __builtin_amdgcn_readlane(cnt_reg, ul); // the count lives in lane ul's register
                        if (base + c > CAP) {
                            base = prune(ws, ul, K, base, seen_ptr, seen_items);
                            pruned = true;
                        }
                        if (h == hh && is_cand)
                            ws.cand[ul * CAP + base + __popc(m & ((1u << row) - 1u))] = make_key(acc[r], (int)item);
                        cnt_reg = lane == ul ? base + c : cnt_reg;
                        lds_fence();
                    }
                }
            }
        }
        if (pruned)
            reload_thr();
    };

    int64_t prev_item = 0; // item of this lane in the block whose accumulators wait for their compares
    bool have_prev = false;
    auto tile_body = [&](auto par_c, int64_t t) __attribute__((always_inline)) { // PAR: which accumulator the tile's first block lands in
        constexpr int PAR = decltype(par_c)::value;
        const int cur = (int)(t & 1);
        const int64_t tile_base = i0 + t * TI;
        if (t + 1 < n_tiles)
            stage_load(tile_base + TI); // in flight while this tile is multiplied

        const float *qb = qt + cur * TI * LDQ;
        // B operands are read a segment (up to 32 MFMAs' worth) ahead into the other half of a register double
        // buffer, so an LDS round trip is exposed once per tile, not once per pair of reads
        constexpr int SEGQ = (HS / 4) < 8 ? (HS / 4) : 8; // float4s per segment
        constexpr int SPB = (HS / 4) / SEGQ;              // segments per block
        constexpr int NSEG = NBLK * SPB;
        float4 bb[2][SEGQ];
        auto load_seg = [&](float4(&dst)[SEGQ], int sgi) __attribute__((always_inline)) {
            const float *bp = qb + ((sgi / SPB) * 32 + row) * LDQ + h * HS + (sgi % SPB) * SEGQ * 4;
#pragma unroll
            for (int q = 0; q < SEGQ; ++q)
                dst[q] = *reinterpret_cast<const float4 *>(bp + 4 * q);
        };
        load_seg(bb[0], 0);
        uint32_t hit = 0;
        bool prev_ok = false;
#pragma unroll
        for (int sgi = 0; sgi < NSEG; ++sgi) {
            const int blk = sgi / SPB, part = sgi % SPB;
            f32x16 &acc = ((PAR + blk) & 1) ? accB : accA;
            const f32x16 &prv = ((PAR + blk) & 1) ? accA : accB;
            if (sgi + 1 < NSEG)
                load_seg(bb[(sgi + 1) & 1], sgi + 1);
            __builtin_amdgcn_sched_barrier(0); // the reads stay ahead of the MFMA chain
            if (part == 0) {
                // accumulator register r of lane (row, h) is score[user (r&3) + 8*(r>>2) + 4*h][item base + row]
                prev_ok = have_prev && prev_item < i1;
                hit = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[r] = 0.f;
            }
#pragma unroll
            for (int qq = 0; qq < SEGQ; ++qq) {
                const int q = part * SEGQ + qq;
                const float4 b = bb[sgi & 1][qq];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 0], b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 1], b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 2], b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 3], b.w, acc, 0, 0, 0);
                // this step's share of the previous block's 16 compares
#pragma unroll
                for (int r = (16 * q) / (HS / 4); r < (16 * (q + 1)) / (HS / 4); ++r)
                    hit |= (prev_ok && prv[r] > thr_r[r]) ? (1u << r) : 0u;
            }
            if (part == SPB - 1) {
                if (__ballot(hit != 0))
                    take_candidates(prv, hit, prev_item);
                prev_item = tile_base + blk * 32 + row;
                have_prev = true;
            }
        }

        if (t + 1 < n_tiles)
            stage_store(cur ^ 1);
        __syncthreads();
    };
    if (NBLK % 2 == 0) {
        for (int64_t t = 0; t < n_tiles; ++t)
            tile_body(std::integral_constant<int, 0>{}, t);
    } else {
        for (int64_t t = 0; t < n_tiles; t += 2) {
            tile_body(std::integral_constant<int, 0>{}, t);
            if (t + 1 < n_tiles)
                tile_body(std::integral_constant<int, 1>{}, t + 1);
        }
    }
    if (have_prev) { // the last block of the sweep still waits for its compares
        const bool inB = ((n_tiles * NBLK - 1) & 1) != 0;
        uint32_t hit = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            hit |= (prev_item < i1 && (inB ? accB[r] : accA[r]) > thr_r[r]) ? (1u << r) : 0u;
        if (__ballot(hit != 0)) {
            if (inB)
                take_candidates(accB, hit, prev_item);
            else
                take_candidates(accA, hit, prev_item);
        }
    }

    // ---- final prune + hand the K best keys of this split to the merge -------------------------
    for (int ul = 0; ul < 32; ++ul) {
        const int gu = blockIdx.x * USERS_PER_WG + wv * 32 + ul;
        if (gu >= Bt)
            break;
        const int n = prune(ws, ul, K, __shfl(cnt_reg, ul, 64), seen_ptr, seen_items);
        if (lane < K)
            part[((int64_t)gu * S + sp) * K + lane] = lane < n ? ws.cand[ul * CAP + lane] : 0;
    }
}

// one wave per user: rank the S*K keys, best first
// (out_stride, out_off: a list longer than 32 is ranked 32 at a time, every pass into its columns of the [Bt, out_stride] outputs)
__global__ __launch_bounds__(64) void eval_merge(const u64 *__restrict__ part, int S, int K, float sign,
                                                 int32_t *__restrict__ topk_ids, float *__restrict__ topk_scores, int out_stride, int out_off)
{
    __shared__ u64 keys[MAX_SPLITS * K_MAX];
    const int lane = threadIdx.x, n = S * K;
    const int64_t user = blockIdx.x;
    for (int k = lane; k < n; k += 64)
        keys[k] = part[user * n + k];
    if (lane < K) {
        topk_ids[user * out_stride + out_off + lane] = -1;
        topk_scores[user * out_stride + out_off + lane] = sign * -INFINITY;
    }
    __syncthreads();
    for (int k = lane; k < n; k += 64) {
        const u64 mine = keys[k];
        if (mine == 0)
            continue;
        int rank = 0;
        for (int t = 0; t < n; ++t)
            rank += keys[t] > mine;
        if (rank < K) {
            topk_ids[user * out_stride + out_off + rank] = key_item(mine);
            topk_scores[user * out_stride + out_off + rank] = sign * key_score(mine);
        }
    }
}

// cal_ranking_metrics (utils/metrics.py:9-19) in float64: one wave per user.  Lanes stride over the user's test
// items: each finds its item's position in the recommended list (or not), and adds its share of hit / 1/(rank+1) /
// 1/log2(rank+2) for every K of the list and of the ideal DCG; the lane sums are combined by a fixed butterfly.
__device__ inline double wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(256) void rank_metrics_kernel(const int32_t *__restrict__ topk_ids,
                                                           const int32_t *__restrict__ users, int Bt, int Kmax,
                                                           const int32_t *__restrict__ ks, int nk,
                                                           const int64_t *__restrict__ test_ptr,
                                                           const int32_t *__restrict__ test_items,
                                                           double *__restrict__ hr, double *__restrict__ mrr,
                                                           double *__restrict__ ndcg)
{
    __shared__ int32_t rec_all[4][K_LIST_MAX];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wv;
    if (r >= Bt)
        return;
    int32_t *rec = rec_all[wv];
    for (int k = lane; k < K_LIST_MAX; k += 64)
        rec[k] = k < Kmax ? topk_ids[r * Kmax + k] : -1; // Kmax <= K_LIST_MAX, checked by the host
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
    const int32_t uu = users[r];
    const int64_t t0 = test_ptr[uu], n_real = test_ptr[uu + 1] - t0;
    double idcg = 0.0;
    for (int64_t id = lane; id < n_real; id += 64)
        idcg += 1.0 / log2((double)(id + 2));
    idcg = wave_sum_f64(idcg);
    for (int q = 0; q < nk; ++q) {
        const int K = ks[q];
        double hit = 0.0, rr = 0.0, dcg = 0.0;
        for (int64_t id = lane; id < n_real; id += 64) {
            const int32_t item = test_items[t0 + id];
            int at = -1; // rec.index(item) within the first K
            for (int idx = K - 1; idx >= 0; --idx)
                at = rec[idx] == item ? idx : at;
            if (at >= 0) {
                hit += 1.0;
                rr += 1.0 / (double)(at + 1);
                dcg += 1.0 / log2((double)(at + 2));
            }
        }
        hit = wave_sum_f64(hit);
        rr = wave_sum_f64(rr);
        dcg = wave_sum_f64(dcg);
        if (lane == 0) {
            const double denom = (double)(K < n_real ? (int64_t)K : n_real);
            hr[(int64_t)q * Bt + r] = hit / denom;
            mrr[(int64_t)q * Bt + r] = rr;
            ndcg[(int64_t)q * Bt + r] = dcg / idcg;
        }
    }
}

// membership bitmap of the users' lists: bits[r][item / 32] |= 1 << (item % 32) for every listed item of user row r
// (one wave per user row; integer OR is order-independent)
__global__ __launch_bounds__(256) void list_bitmap(const int32_t *__restrict__ users, int Bt, const int64_t *__restrict__ ptr,
                                                   const int32_t *__restrict__ items, int words, uint32_t *__restrict__ bits)
{
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= Bt)
        return;
    const int32_t uu = users[r];
    for (int64_t e = ptr[uu] + (threadIdx.x & 63); e < ptr[uu + 1]; e += 64) {
        const int32_t it = items[e];
        atomicOr(bits + r * words + (it >> 5), 1u << (it & 31));
    }
}

// K > 32, bitmap mode: the ids a pass has ranked leave the competition (seen filter: their bit is set like a listed item's;
// candidates mode: cleared)
__global__ __launch_bounds__(256) void ranked_out_of_bitmap(const int32_t *__restrict__ ids, int Bt, int stride, int off, int n, int words,
                                                            int keep_listed, uint32_t *__restrict__ bits)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)Bt * n)
        return;
    const int64_t r = t / n;
    const int32_t it = ids[r * stride + off + (int)(t % n)];
    if (it < 0)
        return;
    if (keep_listed)
        atomicAnd(bits + r * words + (it >> 5), ~(1u << (it & 31)));
    else
        atomicOr(bits + r * words + (it >> 5), 1u << (it & 31));
}

template <int DMAX, int TI, int NW> constexpr size_t sweep_lds()
{
    return sizeof(float) * 2 * TI * (DMAX + 4) + sizeof(u64) * NW * 32 * cap_of(NW) + sizeof(int) * NW * 32 * 4;
}

template <int DMAX, int TI, int NW>
int launch_sweep(crec_ctx *ctx, dim3 grid, hipStream_t st, const float *P, const float *Q, int64_t I, int d,
                 const int32_t *users, int Bt, const int64_t *seen_ptr, const int32_t *seen_items, int K, float sign,
                 int64_t per_split, u64 *part, const uint32_t *bits, int words, const int32_t *excl, int n_excl, int excl_stride)
{
    constexpr size_t lds = sweep_lds<DMAX, TI, NW>();
    static_assert(lds <= 160 * 1024, "eval_sweep shape does not fit the LDS");
    CREC_HIP(ctx, hipFuncSetAttribute((const void *)eval_sweep<DMAX, TI, NW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds));
    crec_prof_begin(ctx, CREC_K_EVAL_SWEEP, st);
    CREC_KLAUNCH(ctx, (eval_sweep<DMAX, TI, NW>), grid, dim3(NW * 64), lds, st, P, Q, I, d, users, Bt, seen_ptr, seen_items, K,
                       sign, per_split, part, bits, words, excl, n_excl, excl_stride);
    crec_prof_end(ctx, st);
    return CREC_OK;
}

} // namespace

static int eval_topk_impl(crec_ctx *ctx, const float *P, const float *Q, int64_t I, int32_t d, const int32_t *users,
                          int32_t Bt, const int64_t *seen_ptr, const int32_t *seen_items, int32_t K, int32_t ascending,
                          int32_t *topk_ids, float *topk_scores, void *stream, int keep_listed);

extern "C" int crec_eval_topk(crec_ctx *ctx, const float *P, const float *Q, int64_t I, int32_t d,
                              const int32_t *users, int32_t Bt, const int64_t *seen_ptr, const int32_t *seen_items,
                              int32_t K, int32_t ascending, int32_t *topk_ids, float *topk_scores, void *stream)
{
    return eval_topk_impl(ctx, P, Q, I, d, users, Bt, seen_ptr, seen_items, K, ascending, topk_ids, topk_scores, stream, 0);
}

extern "C" int crec_eval_topk_candidates(crec_ctx *ctx, const float *P, const float *Q, int64_t I, int32_t d,
                                         const int32_t *users, int32_t Bt, const int64_t *cand_ptr,
                                         const int32_t *cand_items, int32_t K, int32_t ascending, int32_t *topk_ids,
                                         float *topk_scores, void *stream)
{
    return eval_topk_impl(ctx, P, Q, I, d, users, Bt, cand_ptr, cand_items, K, ascending, topk_ids, topk_scores, stream, 1);
}

static int eval_topk_impl(crec_ctx *ctx, const float *P, const float *Q, int64_t I, int32_t d, const int32_t *users,
                          int32_t Bt, const int64_t *seen_ptr, const int32_t *seen_items, int32_t K, int32_t ascending,
                          int32_t *topk_ids, float *topk_scores, void *stream, int keep_listed)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, P && Q && users && seen_ptr && topk_ids && topk_scores, "null argument");
    CREC_REQUIRE(ctx, d >= 4 && d <= 256 && d % 4 == 0, "embed_size %d: need a multiple of 4 in [4, 256]", d);
    CREC_REQUIRE(ctx, K >= 1 && K <= K_LIST_MAX, "K = %d: lists of at most %d ids", K, K_LIST_MAX);
    CREC_REQUIRE(ctx, I >= 1 && I < ((int64_t)1 << 31) && Bt >= 0, "bad sizes");
    CREC_REQUIRE(ctx, (((uintptr_t)P | (uintptr_t)Q) & 15) == 0, "table base not 16-byte aligned");
    if (Bt == 0)
        return CREC_OK;
    const crec_range eval_range(keep_listed ? "crec_eval_topk_candidates" : "crec_eval_topk");
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    // The sweep keeps at most K_MAX = 32 candidates per user on chip.  A longer list (the reference takes any topk,
    // Recommender.py:27) is ranked 32 at a time: pass p ranks the best 32 of what the passes before it have not ranked --
    // their ids join the seen filter (bitmap mode) or are dropped in the prune (list mode) -- into columns [32 p, 32 p + 32).
    const int Kp = K < K_MAX ? K : K_MAX; // buffers and shapes are sized for the widest pass

    // shape: 8 waves per workgroup (two per SIMD) when K and the row width leave room for their buffers (measured:
    // 66 vs 51 % of the MFMA peak at 65536 x 1M x 64, and 0.46 vs 0.67 ms for the ml-1m sweep)
    const int64_t cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    // option eval_shape = 8 / 4: force a shape where it is legal (tests, tuning; crec_ctx_set_option)
    const bool wide = Kp <= cap_of(8) - 32 && d <= 128 && (ctx->knob_eval_shape ? ctx->knob_eval_shape == 8 : Bt > 128);
    const int tile = wide ? (d > 64 ? 32 : 64) : (d > 128 ? 32 : 64);
    const int users_per_wg = (wide ? 8 : 4) * 32;
    const int user_tiles = (int)crec_ceil_div(Bt, users_per_wg);
    const int64_t n_item_tiles = crec_ceil_div(I, tile);
    int64_t S = crec_ceil_div((wide ? 1 : 2) * cus, user_tiles);
    S = S < 1 ? 1 : (S > MAX_SPLITS ? MAX_SPLITS : S);
    S = S > n_item_tiles ? n_item_tiles : S;
    const int64_t per_split = crec_ceil_div(n_item_tiles, S) * tile;
    S = crec_ceil_div(I, per_split);

    // small catalogues: the users' lists as bitmaps, so that a prune tests membership with one load instead of a
    // binary search in global memory (the startup of every (user, split) sweep is prune-bound)
    const int words = (int)crec_ceil_div(I, 32);
    const size_t part_bytes = (sizeof(u64) * (size_t)Bt * S * Kp + 255) / 256 * 256;
    const size_t bits_bytes = sizeof(uint32_t) * (size_t)Bt * words;
    const bool use_bits = bits_bytes <= ((size_t)64 << 20) && !ctx->knob_eval_no_bitmap;
    if (int rc = crec_reserve(ctx, &ctx->eval_ws, part_bytes + (use_bits ? bits_bytes : 0)))
        return rc;
    u64 *part = (u64 *)ctx->eval_ws.ptr;
    uint32_t *bits = use_bits ? (uint32_t *)((char *)ctx->eval_ws.ptr + part_bytes) : nullptr;
    if (use_bits) {
        CREC_HIP(ctx, hipMemsetAsync(bits, 0, bits_bytes, st));
        hipLaunchKernelGGL(list_bitmap, dim3((unsigned)crec_ceil_div(Bt, 4)), dim3(256), 0, st, users, (int)Bt, seen_ptr, seen_items,
                           words, bits);
    }
    const float sign = ascending ? -1.0f : 1.0f;
    dim3 grid((unsigned)user_tiles, (unsigned)S);
    for (int done = 0; done < K; done += K_MAX) {
        const int kpass = (K - done) < K_MAX ? (K - done) : K_MAX;
        const int kk = kpass | (keep_listed << 16);
        const int32_t *excl = use_bits ? nullptr : topk_ids; // list mode: the columns ranked so far, read by the prune
        const int n_excl = use_bits ? 0 : done;
        int rc;
#define CREC_SWEEP(DM, TL, W) \
    launch_sweep<DM, TL, W>(ctx, grid, st, P, Q, I, d, users, Bt, seen_ptr, seen_items, kk, sign, per_split, part, bits, words, excl, n_excl, K)
        if (wide)
            rc = d <= 32 ? CREC_SWEEP(32, 64, 8) : d <= 64 ? CREC_SWEEP(64, 64, 8) : CREC_SWEEP(128, 32, 8);
        else
            rc = d <= 32 ? CREC_SWEEP(32, 64, 4) : d <= 64 ? CREC_SWEEP(64, 64, 4) : d <= 128 ? CREC_SWEEP(128, 64, 4) : CREC_SWEEP(256, 32, 4);
#undef CREC_SWEEP
        if (rc)
            return rc;
        crec_prof_begin(ctx, CREC_K_EVAL_MERGE, st);
        CREC_KLAUNCH(ctx, eval_merge, dim3((unsigned)Bt), dim3(64), 0, st, (const u64 *)part, (int)S, kpass, sign, topk_ids, topk_scores, (int)K, done);
        crec_prof_end(ctx, st);
        if (use_bits && done + K_MAX < K)
            hipLaunchKernelGGL(ranked_out_of_bitmap, dim3((unsigned)crec_ceil_div((int64_t)Bt * kpass, 256)), dim3(256), 0, st, (const int32_t *)topk_ids,
                               (int)Bt, (int)K, done, kpass, words, keep_listed, bits);
    }
    CREC_HIP(ctx, hipGetLastError());
    return crec_prof_flush(ctx, st);
}

extern "C" int crec_rank_metrics(crec_ctx *ctx, const int32_t *topk_ids, const int32_t *users, int32_t Bt, int32_t K,
                                 const int32_t *ks, int32_t nk, const int64_t *test_ptr, const int32_t *test_items,
                                 double *hr, double *mrr, double *ndcg, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, topk_ids && users && ks && test_ptr && hr && mrr && ndcg, "null argument");
    CREC_REQUIRE(ctx, nk >= 1 && nk <= 64 && Bt >= 0 && K >= 1 && K <= K_LIST_MAX, "bad sizes (lists of at most %d ids)", K_LIST_MAX);
    for (int q = 0; q < nk; ++q)
        CREC_REQUIRE(ctx, ks[q] >= 1 && ks[q] <= K, "ks[%d] = %d outside [1, K]", q, ks[q]);
    if (Bt == 0)
        return CREC_OK;
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    if (int rc = crec_reserve(ctx, &ctx->ks_dev, sizeof(int32_t) * 64))
        return rc;
    CREC_HIP(ctx, hipMemcpyAsync(ctx->ks_dev.ptr, ks, sizeof(int32_t) * nk, hipMemcpyHostToDevice, st));
    crec_prof_begin(ctx, CREC_K_METRICS, st);
    CREC_KLAUNCH(ctx, rank_metrics_kernel, dim3((unsigned)crec_ceil_div(Bt, 4)), dim3(256), 0, st, topk_ids, users,
                       Bt, K, (const int32_t *)ctx->ks_dev.ptr, nk, test_ptr, test_items, hr, mrr, ndcg);
    crec_prof_end(ctx, st);
    CREC_HIP(ctx, hipGetLastError());
    return crec_prof_flush(ctx, st);
}
