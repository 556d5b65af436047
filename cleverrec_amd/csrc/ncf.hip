// Pointwise NCF family for gfx950: GMF, MLP and NeuMF training steps (sigmoid cross-entropy on
// (u, i, y) instances from pointwise_ranking_sampler).
//
// Replaces sess.run([train, loss], {u_idx, i_idx, y}) (model/RankingRecommender.py:52-58) on the graphs
// of model/ranking/GMF.py:29-57, MLP.py:31-77, NeuMF.py:33-105; see include/crec.h (crec_ncf).
//
// Per batch, two launches:
//   ncf_tile        one workgroup (8 waves) per tile of T = 32 instances, everything on chip: every tower weight,
//                   bias and h staged in LDS once; gathers of the embedding rows; ReLU tower forward and backward
//                   as 16x16x4 fp32 MFMA products between LDS buffers (tile_gemm, ncf_common.h); GMF product, logit,
//                   e = sigmoid(logit) - y, loss term.  Writes per-instance contribution rows for the embedding
//                   tables (dL/drow without the l2 term; layer 0's dL/dX goes straight out) and per-tile partial
//                   gradients of the small dense parameters (W_k, b_k, h) -- all of them written THROUGH the L2
//                   (crec_store.h): left dirty there they were written back at the kernel's end, 4 us per launch.
//   crec_apply_all  (rows_adam.hip) every parameter update in one launch: per embedding table, contribution rows
//                   summed per table row in batch order (+ reg * row per occurrence), TF1 sparse Adam over every
//                   row; the small parameters: tile partials added in 16 fixed slices, TF1 dense ApplyAdam.
// Everything is deterministic (fixed summation orders, no atomics).
#include "crec_internal.h"
#include "crec_store.h"
#include "rows_adam.h"
#include "ncf_common.h"

namespace {

#ifndef NCF_T
#define NCF_T 32
#endif
constexpr int T = NCF_T;     // instances per tile
#ifndef NCF_NT
#define NCF_NT 512
#endif
constexpr int NT = NCF_NT;   // threads per workgroup
constexpr int NW = NT / 64;  // waves per workgroup
constexpr int RW = T / NW;   // tile rows owned by one wave in the row-wise phases
#ifndef NCF_GRID_DIV
#define NCF_GRID_DIV 1 // tiles per workgroup (grid-stride): timing experiments
#endif

#ifndef NCF_DH_LATE
#define NCF_DH_LATE 1 // dL/dh reductions beside the last layer's dW product instead of in front of the backward pass
#endif
#ifndef NCF_DPP
#define NCF_DPP 1 // the logit phase's sums over an instance's 16 lanes as DPP row operations
#endif

__device__ inline float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1)
        v += __shfl_xor(v, off, 64);
    return v;
}

// dynamic LDS (every 2-D array has its row stride padded by one float, so that the MFMA operand reads -- 16 rows at
// one column -- spread over the banks): activations A_0..A_L (A_0 = concat(pm, qm)), deltas D_1..D_L, the gmf rows,
// e / label / (u, i) per instance, the small vectors [b_0 | .. | b_{L-1} | h], and the weights of every layer.
// Nothing in the phases below reads a parameter from global memory: with one workgroup per CU every such load
// would be an exposed round trip.
//
// Shape-specialised: <L0C, LC, DGC> = (layers[0], number of layers, GMF width) fixed at compile time, or L0C < 0 = read
// from NcfDev.  With one 32-instance tile per workgroup every phase's code runs ONCE per launch, and the generic
// kernel spends most of a tile on bookkeeping around its few MFMAs -- offsets of the LDS arrays, block -> (row, column)
// divisions, loop control (stamped: forward pass 10.4 k cycles, 8.5 k of them with the MFMAs removed).  With the
// shape fixed all of that folds into immediates.
template <int LC, class F> __device__ __forceinline__ void layers_up(int L, F f)
{
    if constexpr (LC >= 0) {
        if constexpr (LC > 0) f(0);
        if constexpr (LC > 1) f(1);
        if constexpr (LC > 2) f(2);
        if constexpr (LC > 3) f(3);
    } else {
        for (int k = 0; k < L; ++k)
            f(k);
    }
}
template <int LC, class F> __device__ __forceinline__ void layers_down(int L, F f)
{
    if constexpr (LC >= 0) {
        if constexpr (LC > 3) f(3);
        if constexpr (LC > 2) f(2);
        if constexpr (LC > 1) f(1);
        if constexpr (LC > 0) f(0);
    } else {
        for (int k = L - 1; k >= 0; --k)
            f(k);
    }
}

#ifdef NCF_WPE // timing experiments: cap the registers for NCF_WPE waves per SIMD (two 16-instance tiles per CU)
#define NCF_OCC __attribute__((amdgpu_waves_per_eu(NCF_WPE, NCF_WPE)))
#else
#define NCF_OCC
#endif
template <int L0C, int LC, int DGC>
__global__ __launch_bounds__(NT) NCF_OCC void ncf_tile(NcfDev m, const int32_t *__restrict__ u, const int32_t *__restrict__ i,
                                               const float *__restrict__ y, int nb, float *__restrict__ c_pg,
                                               float *__restrict__ c_qg, float *__restrict__ c_pm,
                                               float *__restrict__ c_qm, float *__restrict__ loss_inst,
                                               float *__restrict__ partial)
{
    constexpr bool FIXED = L0C >= 0;
    constexpr int LCC = FIXED ? LC : -1;
    extern __shared__ float lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63; // wave: scalar
#ifdef NCF_STAMPS // debug builds: core-clock and 100 MHz stamps of tile 7 at the phase boundaries
    int stamp_n = 0;
#define STAMP()                                                          \
    do {                                                                 \
        __syncthreads();                                                 \
        if (m.dbg && blockIdx.x == 7 && tid == 0) {                          \
            m.dbg[2 * stamp_n] = __builtin_amdgcn_s_memtime();           \
            m.dbg[2 * stamp_n + 1] = __builtin_amdgcn_s_memrealtime();   \
        }                                                                \
        ++stamp_n;                                                       \
    } while (0)
#else
#define STAMP()
#endif
    for (int tile = blockIdx.x; tile * T < nb; tile += gridDim.x) { // grid-stride over the batch's tiles
    STAMP();
    const int r0 = tile * T;
    const int nr = (nb - r0) < T ? (nb - r0) : T;
    const bool has_g = FIXED ? DGC > 0 : m.kind != 1, has_m = FIXED ? LC > 0 : m.kind != 0;
    const int L0 = FIXED ? L0C : m.L0;
    const int dg = FIXED ? DGC : (has_g ? m.dg : 0), L = FIXED ? LC : (has_m ? m.n_layers : 0), hm = L0 / 2, ldg = dg + 1;

    // carve LDS (ncf_tile_lds_floats below mirrors this).  Widths halve exactly (ncf_describe), so the offsets
    // are closed-form scalar arithmetic: no pointer tables (they would live in scratch)
    const auto width_of = [&](int k) { return has_m ? (L0 >> k) : 0; };
    const auto off_of = [&](int k) { // floats before A_k when the A_j are laid out back to back
        int o = 0;
        for (int j = 0; j < k; ++j)
            o += T * (width_of(j) + 1);
        return o;
    };
    const auto woff_of = [&](int k) { // floats before W_k (rows padded by one float)
        int o = 0;
        for (int j = 0; j < k; ++j)
            o += width_of(j) * (width_of(j + 1) + 1);
        return o;
    };
    const auto boff_of = [&](int k) { // floats before b_k
        int o = 0;
        for (int j = 0; j < k; ++j)
            o += width_of(j + 1);
        return o;
    };
    const auto fw_of = [&](int k) { // flat offset of W_k in the parameter block [W_0 | b_0 | ... | h]; fw_of(L) = h's
        int o = 0;
        for (int j = 0; j < k; ++j)
            o += width_of(j) * width_of(j + 1) + width_of(j + 1);
        return o;
    };
    const auto fb_of = [&](int k) { return fw_of(k) + width_of(k) * width_of(k + 1); };
    const int h_off = fw_of(L);
    const int last = has_m ? width_of(L) : 0;
    float *const A_base = lds;
    float *const D_base = lds + off_of(L + 1) - off_of(1); // D_k sits where an (L+1+k)-th activation would
    float *p = lds + 2 * off_of(L + 1) - off_of(1);
#define A_(k) (A_base + off_of(k))
#define D_(k) (D_base + off_of(k))
#define W_(k) width_of(k)
#define LD_(k) (width_of(k) + 1)
    float *const G_p = p;
    p += T * ldg;
    float *const G_q = p;
    p += T * ldg;
    float *const e = p, *const yl = p + T;
    float *const b_base = p + 4 * T; // (2T floats free: the layout is shared with ncf_tile_lds_floats)
    float *const hl = b_base + boff_of(L); // GMF: h_gmf[dg]; MLP: h_mlp[last]; NeuMF: h_neumf[dg + last]
    float *const W_base = hl + dg + last;
#define WL_(k) (W_base + woff_of(k))
#define BL_(k) (b_base + boff_of(k))

    // ---- prologue: everything the tile needs from memory in TWO dependent round trips (indices -> rows), the parameter
    // block travelling beside them.  (One load loop per layer and vector, each followed by its LDS writes, was seven
    // round trips back to back: 4.3 us of a 26 us tile.)
    const float yv = tid < nr ? y[r0 + (tid < T ? tid : 0)] : 0.f; // (threads past T: a valid address, value unused)
    // a wave owns RW rows of the tile, lanes run along the row.  Their indices: ONE vector load per column (lane l asks for
    // row l % RW; a branch-free clamp keeps ragged tiles in bounds), handed round as scalars -- wave-uniform scalar loads
    // would each wait for the one before (stamped: eight dependent trips)
    const int rl = wave * RW + (lane % RW), rc = rl < nr ? rl : nr - 1;
    const int myu = u[r0 + rc], myi = i[r0 + rc];
    // the flat parameter block [W_0 | b_0 | ... | h] -> LDS (weight rows padded by one float): where flat offset o lands
    // (float index into lds); the walk over the layers is scalar arithmetic, only the compares are per lane
    const int W_lds = (int)(W_base - lds), b_lds = (int)(b_base - lds), h_lds = (int)(hl - lds);
    const auto dst_of = [&](int o) {
        int wo = 0, lw = W_lds, lb = b_lds, w = has_m ? L0 : 0, dst = -1;
        for (int k = 0; k < L; ++k) {
            const int kout = w >> 1, nW = w * kout, idx = o - wo;
            const int row = (kout & (kout - 1)) == 0 ? idx >> __builtin_ctz(kout) : idx / kout;
            const int dW = lw + row * (kout + 1) + (idx - row * kout), dB = lb + (idx - nW);
            dst = dst < 0 && idx < nW ? dW : dst;
            dst = dst < 0 && idx < nW + kout ? dB : dst;
            wo += nW + kout, lw += w * (kout + 1), lb += kout, w = kout;
        }
        return dst < 0 ? h_lds + (o - wo) : dst;
    };
    bool quads = (dg & 3) == 0; // a float4 of the block never straddles a weight row or a vector
    for (int k = 0; k < L; ++k)
        quads = quads && (W_(k + 1) & 3) == 0;
    constexpr int SD = 8; // float4 loads of the parameter block in flight per thread
    const bool vec = ((uintptr_t)m.small & 15) == 0;
    const int n_small = FIXED ? h_off + dg + last : m.n_small;
    const int n4 = vec && quads ? n_small >> 2 : 0;
    float4 w4[SD];
#pragma unroll
    for (int e = 0; e < SD; ++e) {
        const int o4 = e * NT + tid;
        w4[e] = ((const float4 *)m.small)[o4 < n4 ? o4 : 0]; // never predicated: a skipped load costs the ones behind it their place in the queue (vmcnt)
    }
    __builtin_amdgcn_sched_barrier(0);
    int64_t ub[RW], ib[RW];
    bool ok[RW];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        ok[rr] = wave * RW + rr < nr;
        ub[rr] = __builtin_amdgcn_readlane(myu, rr);
        ib[rr] = __builtin_amdgcn_readlane(myi, rr);
    }
    // first 64 columns of every gathered row: asked for as soon as the indices are back, while the parameters travel
    float gp[RW], gq[RW], ap[RW], aq[RW];
    // (rows past the batch's end repeat its last row and lanes past a narrow row its last column: always a valid
    // address, never a branch; what they load is either not stored or multiplied by e = 0 in the backward pass)
    const int cg = lane < dg ? lane : dg - 1, cm = lane < hm ? lane : hm - 1;
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
        if (has_g) {
            gp[rr] = m.Pg[ub[rr] * dg + cg];
            gq[rr] = m.Qg[ib[rr] * dg + cg];
        }
        if (has_m) {
            ap[rr] = m.Pm[ub[rr] * hm + cm];
            aq[rr] = m.Qm[ib[rr] * hm + cm];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (tid < T)
        yl[tid] = yv;
#pragma unroll
    for (int e = 0; e < SD; ++e) {
        const int o4 = e * NT + tid;
        if (o4 < n4) {
            float *dst = lds + dst_of(4 * o4);
            dst[0] = w4[e].x, dst[1] = w4[e].y, dst[2] = w4[e].z, dst[3] = w4[e].w;
        }
    }
    for (int o4 = SD * NT + tid; o4 < n4; o4 += NT) { // wider towers: the rest of the block
        const float4 w = ((const float4 *)m.small)[o4];
        float *dst = lds + dst_of(4 * o4);
        dst[0] = w.x, dst[1] = w.y, dst[2] = w.z, dst[3] = w.w;
    }
    for (int o = 4 * n4 + tid; o < n_small; o += NT) // odd shapes / the block's last floats
        lds[dst_of(o)] = m.small[o];
    STAMP(); // 1: parameters staged
    {
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int r = wave * RW + rr;
            if (lane < dg) {
                G_p[r * ldg + lane] = gp[rr];
                G_q[r * ldg + lane] = gq[rr];
            }
            if (has_m && lane < hm) {
                A_(0)[r * LD_(0) + lane] = ap[rr];
                A_(0)[r * LD_(0) + hm + lane] = aq[rr];
            }
        }
        for (int c = 64 + lane; c < dg; c += 64) { // rows wider than a wave
            float tp[RW], tq[RW];
#pragma unroll
            for (int rr = 0; rr < RW; ++rr) {
                tp[rr] = ok[rr] ? m.Pg[ub[rr] * dg + c] : 0.f;
                tq[rr] = ok[rr] ? m.Qg[ib[rr] * dg + c] : 0.f;
            }
#pragma unroll
            for (int rr = 0; rr < RW; ++rr) {
                G_p[(wave * RW + rr) * ldg + c] = tp[rr];
                G_q[(wave * RW + rr) * ldg + c] = tq[rr];
            }
        }
        if (has_m) {
            float *A0 = A_(0);
            const int ld0 = LD_(0);
            for (int c = 64 + lane; c < hm; c += 64) {
                float tp[RW], tq[RW];
#pragma unroll
                for (int rr = 0; rr < RW; ++rr) {
                    tp[rr] = ok[rr] ? m.Pm[ub[rr] * hm + c] : 0.f;
                    tq[rr] = ok[rr] ? m.Qm[ib[rr] * hm + c] : 0.f;
                }
#pragma unroll
                for (int rr = 0; rr < RW; ++rr) {
                    A0[(wave * RW + rr) * ld0 + c] = tp[rr];
                    A0[(wave * RW + rr) * ld0 + hm + c] = tq[rr];
                }
            }
        }
    }
    STAMP(); // 2: rows gathered
    // ---- tower forward: A[k+1] = relu(A[k] W_k + b_k) --------------------------------------------------------
    layers_up<LCC>(L, [&](int k) __attribute__((always_inline)) {
        const int kin = W_(k), kout = W_(k + 1), ws = kout + 1, lda = LD_(k), ldo = LD_(k + 1);
        const float *b = BL_(k), *Ak = A_(k), *Wl = WL_(k);
        float *An = A_(k + 1);
        __syncthreads(); // A[k] complete
#if NCF_ABL == 5 // per-wave trace of the forward layers of tile 7: entry and exit of the product
        if (m.dbg && blockIdx.x == 7 && lane == 0)
            m.dbg[128 + wave * 16 + 2 * k] = __builtin_amdgcn_s_memtime();
#endif
        const auto fa_f = [&](int r, int kk) { return Ak[r * lda + kk]; };
        const auto fb_f = [&](int kk, int c) { return Wl[kk * ws + c]; };
        const auto fs_f = [&](int r, int c, float s) {
            s += b[c];
            An[r * ldo + c] = s > 0.f ? s : 0.f;
        };
        tile_gemm<NW>(T, kout, kin, fa_f, fb_f, fs_f);
#if NCF_ABL == 5
        if (m.dbg && blockIdx.x == 7 && lane == 0)
            m.dbg[128 + wave * 16 + 2 * k + 1] = __builtin_amdgcn_s_memtime();
#endif
#if NCF_ABL == 4
        STAMP(); // per layer
#endif
    });
    __syncthreads();
    STAMP(); // 3..: forward
    // ---- logit, e, loss: LPI lanes per instance (every thread of the workgroup), each over a contiguous share of the
    // columns, four reads in flight, then a fixed butterfly over the LPI -----------------------------------------------
    constexpr int LPI = NT / T;
    static_assert(LPI >= 1 && (LPI & (LPI - 1)) == 0 && LPI <= 64, "threads per instance");
    {
        const int r = tid / LPI, sub = tid % LPI;
        const auto share = [&](int n, int &c0, int &c1) {
            const int cw = (n + LPI - 1) / LPI;
            c0 = sub * cw < n ? sub * cw : n;
            c1 = c0 + cw < n ? c0 + cw : n;
        };
        float zg = 0.f, ng = 0.f, zm = 0.f, nm = 0.f;
        int c0, c1;
        share(dg, c0, c1);
#pragma unroll 4
        for (int c = c0; c < c1; ++c) {
            const float pg = G_p[r * ldg + c], qg = G_q[r * ldg + c];
            zg = zg + (pg * qg) * hl[c];
            ng = ng + (pg * pg + qg * qg);
        }
        if (has_m) {
            const float *AL = A_(L), *A0 = A_(0);
            const int ldl = LD_(L), ld0 = LD_(0);
            share(last, c0, c1);
#pragma unroll 4
            for (int c = c0; c < c1; ++c)
                zm = zm + AL[r * ldl + c] * hl[dg + c];
            share(W_(0), c0, c1);
#pragma unroll 8
            for (int c = c0; c < c1; ++c) {
                const float a = A0[r * ld0 + c];
                nm = nm + a * a;
            }
        }
        if constexpr (LPI == 16 && NCF_DPP) {
            zg = row16_sum(zg), ng = row16_sum(ng), zm = row16_sum(zm), nm = row16_sum(nm);
        } else {
#pragma unroll
            for (int off = 1; off < LPI; off <<= 1) {
                zg += __shfl_xor(zg, off, 64);
                ng += __shfl_xor(ng, off, 64);
                zm += __shfl_xor(zm, off, 64);
                nm += __shfl_xor(nm, off, 64);
            }
        }
        if (sub == 0) {
            const bool ok = r < nr;
            const float z = zg + zm, lab = yl[r];
            // d loss / d logit: sigmoid(z) - y for the cross-entropy, 2 (z - y) for the square loss (utils/tools.py:74-75)
            e[r] = !ok ? 0.f : (m.loss == CREC_LOSS_SQUARE ? 2.0f * (z - lab) : sigmoidf_(z) - lab);
            if (ok) {
                // sigmoid_cross_entropy_with_logits (utils/tools.py:68-69) | squared_difference(y, logits) (:74-75)
                // + the l2 terms of the gathered rows
                const float ls = m.loss == CREC_LOSS_SQUARE ? (lab - z) * (lab - z)
                                                            : (z > 0.f ? z : 0.f) - z * lab + log1pf(expf(-fabsf(z)));
                loss_inst[r0 + r] = ls + m.reg_g * (0.5f * ng) + m.reg_m * (0.5f * nm);
            }
        }
    }
    __syncthreads();
    float *part = partial + (int64_t)tile * n_small;
    STAMP(); // logit
    // dL/dh_g (rows past nr carry e = 0): the LAST wave's job, dL/dh_m the one before's -- side by side.  With a tower they run
    // beside the last layer's dW product (two or four blocks: most waves have none), not in front of the backward pass.
    const auto dh_g_job = [&]() __attribute__((always_inline)) {
        for (int c = wave == NW - 1 ? lane : dg; c < dg; c += 64) {
            float s = 0.f;
#pragma unroll 16
            for (int r = 0; r < T; ++r)
                s = s + e[r] * (G_p[r * ldg + c] * G_q[r * ldg + c]);
            st_out(part + h_off + c, s);
        }
    };
    const auto dh_m_job = [&]() __attribute__((always_inline)) {
        const float *AL = A_(L);
        const int ldl = LD_(L);
        for (int c = wave == (NW >= 2 ? NW - 2 : 0) ? lane : last; c < last; c += 64) {
            float s = 0.f;
#pragma unroll 16
            for (int r = 0; r < T; ++r)
                s = s + e[r] * AL[r * ldl + c];
            st_out(part + h_off + dg + c, s);
        }
    };
    // ---- GMF backward ----------------------------------------------------------------------------------------
    if (has_g) {
        for (int c = lane; c < dg; c += 64) {
            const float hc = hl[c];
#pragma unroll
            for (int rr = 0; rr < RW; ++rr) {
                const int r = wave * RW + rr;
                if (r < nr) {
                    const float er = e[r];
                    st_out(c_pg + (int64_t)(r0 + r) * dg + c, er * (G_q[r * ldg + c] * hc));
                    st_out(c_qg + (int64_t)(r0 + r) * dg + c, er * (G_p[r * ldg + c] * hc));
                }
            }
        }
        if (!has_m || !NCF_DH_LATE)
            dh_g_job();
    }
    if (has_m) {
        const float *AL = A_(L);
        float *DL = D_(L);
        const int ldl = LD_(L);
        if (!NCF_DH_LATE)
            dh_m_job();
        // D[L] = e * h_m * relu'
        for (int c = lane; c < last; c += 64) {
            const float hc = hl[dg + c];
#pragma unroll
            for (int rr = 0; rr < RW; ++rr) {
                const int r = wave * RW + rr;
                DL[r * ldl + c] = AL[r * ldl + c] > 0.f ? e[r] * hc : 0.f;
            }
        }
        STAMP(); // gmf backward, D[L]
        layers_down<LCC>(L, [&](int k) __attribute__((always_inline)) {
            const int kin = W_(k), kout = W_(k + 1), ws = kout + 1, lda = LD_(k), ldd = LD_(k + 1);
            const float *Ak = A_(k), *Dn = D_(k + 1), *Wl = WL_(k);
            float *Dk = D_(k);
            float *pw = part + fw_of(k);
            __syncthreads(); // D[k+1] complete
            if (NCF_DH_LATE && k == L - 1) {
                if (has_g)
                    dh_g_job();
                dh_m_job();
            }
            // dW_k[kk][c] = sum_r A[k][r][kk] * D[k+1][r][c]   (rows past nr carry D = 0)
            const auto fa_w = [&](int kk, int r) { return Ak[r * lda + kk]; };
            const auto fb_w = [&](int r, int c) { return Dn[r * ldd + c]; };
            const auto fs_w = [&](int kk, int c, float s) { st_out(pw + kk * kout + c, s); };
            tile_gemm<NW>(kin, kout, T, fa_w, fb_w, fs_w);
            for (int c = tid; c < kout; c += NT) { // db_k[c] = sum_r D[k+1][r][c]
                float s = 0.f;
#pragma unroll 16
                for (int r = 0; r < T; ++r)
                    s = s + Dn[r * ldd + c];
                st_out(part + fb_of(k) + c, s);
            }
            STAMP(); // dW_k
            // delta of the layer below: through its ReLU into D[k] (k > 0), or dL/dX = the contribution rows of the
            // two MLP tables straight to HBM (k == 0)
            if (k > 0)
                tile_gemm<NW>(
                    T, kin, kout, [&](int r, int c) { return Dn[r * ldd + c]; }, [&](int c, int kk) { return Wl[kk * ws + c]; },
                    [&](int r, int kk, float s) {
                        const int o = r * lda + kk;
                        Dk[o] = Ak[o] > 0.f ? s : 0.f;
                    });
            else {
                const auto fa_x = [&](int r, int c) { return Dn[r * ldd + c]; };
                const auto fb_x = [&](int c, int kk) { return Wl[kk * ws + c]; };
                const auto fs_x = [&](int r, int kk, float s) {
                    if (r < nr) {
                        if (kk < hm)
                            st_out(c_pm + (int64_t)(r0 + r) * hm + kk, s);
                        else
                            st_out(c_qm + (int64_t)(r0 + r) * hm + (kk - hm), s);
                    }
                };
                tile_gemm<NW>(T, kin, kout, fa_x, fb_x, fs_x);
            }
            STAMP(); // dX_k
        });
    }
    __syncthreads(); // LDS is reused by the next tile
    }
}

#undef A_
#undef D_
#undef W_
#undef LD_
#undef WL_
#undef BL_
#undef STAMP

size_t ncf_tile_lds_floats(bool has_g, bool has_m, int dg, int L0, int n_layers)
{
    size_t f = 0;
    int w = has_m ? L0 : 0;
    f += (size_t)T * (w + 1);
    for (int k = 0; k < (has_m ? n_layers : 0); ++k) {
        f += (size_t)w * (w / 2 + 1); // W_k, rows padded by one float
        w /= 2;
        f += 2 * (size_t)T * (w + 1) + w; // A_{k+1}, D_{k+1}, b_k
    }
    f += 2 * (size_t)T * ((has_g ? dg : 0) + 1) + 4 * T; // gmf rows; e, label, u, i
    f += (has_g ? dg : 0) + w;                            // h
    return f;
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
    // a batch's slots are indexed 16 384 at a time (one workgroup's sort in LDS); a larger batch in several such chunks, which the
    // update launch walks one after the other -- still every row's slots in ascending order
    CREC_REQUIRE(ctx, B >= 1 && B <= 262144, "batch_size %d: at most 262144", B);
    const int ix_chunks = (int)crec_ceil_div(B, 16384);
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
    const size_t ixu = crec_rows_index_bytes(nb, B, U, ix_chunks), ixi = crec_rows_index_bytes(nb, B, I, ix_chunks);
    const size_t cb = sizeof(float) * (size_t)B * (2 * d.dg + 2 * hm);
    const size_t pb = sizeof(float) * (size_t)n_tiles * d.n_small, lb = sizeof(float) * (size_t)nb * B;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    if (int rc = crec_reserve(ctx, &ctx->idx_ws, up(ixu) + up(ixi) + up(cb) + up(pb) + up(lb)))
        return rc;
    char *ws = (char *)ctx->idx_ws.ptr;
    crec_rows_index_view vu, vi;
    if (int rc = crec_rows_index(ctx, u, N, B, U, ws, &vu, st, ix_chunks))
        return rc;
    if (int rc = crec_rows_index(ctx, i, N, B, I, ws + up(ixu), &vi, st, ix_chunks))
        return rc;
    float *c_pg = (float *)(ws + up(ixu) + up(ixi));
    float *c_qg = c_pg + (size_t)B * d.dg, *c_pm = c_qg + (size_t)B * d.dg, *c_qm = c_pm + (size_t)B * hm;
    float *partial = (float *)(ws + up(ixu) + up(ixi) + up(cb));
    float *loss_inst = (float *)(ws + up(ixu) + up(ixi) + up(cb) + up(pb));

    // LDS of the tile kernel
    const size_t lds_f = ncf_tile_lds_floats(has_g, has_m, d.dg, d.L0, d.n_layers);
    const size_t lds = lds_f * sizeof(float);
    CREC_REQUIRE(ctx, lds <= 160 * 1024, "model too wide for the on-chip tile (%zu bytes of LDS)", lds);
    // the shapes of the shipped confs and of BASELINE configs[2] (layers=[128,64,32], embed_size 32 / 64, NeuMF and MLP)
    // run shape-specialised kernels; everything else the generic one
    using TileFn = void (*)(NcfDev, const int32_t *, const int32_t *, const float *, int, float *, float *, float *, float *, float *,
                            float *);
    TileFn tile_fn = ncf_tile<-1, 0, 0>;
    if (has_m && d.L0 == 128 && d.n_layers == 3) {
        if (!has_g)
            tile_fn = ncf_tile<128, 3, 0>;
        else if (d.dg == 64)
            tile_fn = ncf_tile<128, 3, 64>;
        else if (d.dg == 32)
            tile_fn = ncf_tile<128, 3, 32>;
    }
    if (ctx->knob_ncf_generic)
        tile_fn = ncf_tile<-1, 0, 0>;
    CREC_HIP(ctx, hipFuncSetAttribute((const void *)tile_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

#ifdef NCF_STAMPS
    long long *dbg = nullptr;
    CREC_HIP(ctx, hipMalloc(&dbg, 512 * sizeof(long long)));
    CREC_HIP(ctx, hipMemset(dbg, 0, 512 * sizeof(long long)));
    d.dbg = dbg;
#endif
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t t0 = b * B;
        const int nbt = (int)((N - t0) < B ? (N - t0) : B);
        crec_rows_opt o{};
        o.kind = ROWS_SPARSE_TF1;
        if (int rc = crec_rows_optimizer(ctx, opt, &o, true))
            return rc;
        crec_prof_begin(ctx, CREC_K_NCF_TILE, st);
        CREC_KLAUNCH(ctx, tile_fn, dim3((unsigned)crec_ceil_div(crec_ceil_div(nbt, T), NCF_GRID_DIV)), dim3(NT), lds, st, d, u + t0, i + t0, y + t0, nbt,
                           c_pg, c_qg, c_pm, c_qm, loss_inst + t0, partial);
        crec_prof_end(ctx, st);
        // every parameter update of the step in one launch when the tables share a row-width class
        crec_apply_set set{};
        const auto job = [&](const crec_param &t, int64_t rows, int cols, const crec_rows_index_view &v, const float *c, float reg) {
            crec_rows_opt oo = o;
            oo.reg_occ = reg;
            set.tables[set.n_tables++] = crec_rows_job{crec_rows_table{t.var, t.m, t.v, rows, cols}, v.chunks, v.rowseg + b * v.chunks * v.rows,
                                                       v.order + b * v.per_batch, c, nullptr, nullptr, oo};
        };
        const auto cls = [](int dd) {
            int l = 1;
            while (l * 4 < dd)
                l <<= 1;
            return l;
        };
        crec_rows_opt od = o;
        od.kind = ROWS_DENSE;
        const crec_small_job small{mdl->small.var, mdl->small.m, mdl->small.v, partial, (int)crec_ceil_div(nbt, T), d.n_small,
                                   d.n_small, od};
        if (has_g && has_m && d.dg == hm && o.opt == 0 && o.kind == ROWS_SPARSE_TF1 && !ctx->knob_ncf_generic) {
            // NeuMF with equal widths: a user's P_gmf and P_mlp rows (an item's Q_gmf and Q_mlp rows) share their slots --
            // one wave updates both (rows_apply_row_pair): half the waves, the index walked once
            job(mdl->P_gmf, U, d.dg, vu, c_pg, mdl->reg_gmf);
            set.tables[set.n_tables - 1].t2 = crec_rows_table{mdl->P_mlp.var, mdl->P_mlp.m, mdl->P_mlp.v, U, hm};
            set.tables[set.n_tables - 1].contrib2 = c_pm;
            set.tables[set.n_tables - 1].reg_occ2 = mdl->reg_mlp;
            job(mdl->Q_gmf, I, d.dg, vi, c_qg, mdl->reg_gmf);
            set.tables[set.n_tables - 1].t2 = crec_rows_table{mdl->Q_mlp.var, mdl->Q_mlp.m, mdl->Q_mlp.v, I, hm};
            set.tables[set.n_tables - 1].contrib2 = c_qm;
            set.tables[set.n_tables - 1].reg_occ2 = mdl->reg_mlp;
        } else {
            if (has_g) {
                job(mdl->P_gmf, U, d.dg, vu, c_pg, mdl->reg_gmf);
                job(mdl->Q_gmf, I, d.dg, vi, c_qg, mdl->reg_gmf);
            }
            if (has_g && has_m && cls(d.dg) != cls(hm)) { // two width classes: two launches
                if (int rc = crec_apply_all(ctx, set, st))
                    return rc;
                set = crec_apply_set{};
            }
            if (has_m) {
                job(mdl->P_mlp, U, hm, vu, c_pm, mdl->reg_mlp);
                job(mdl->Q_mlp, I, hm, vi, c_qm, mdl->reg_mlp);
            }
        }
        set.small = small;
        if (int rc = crec_apply_all(ctx, set, st))
            return rc;
        opt->beta1_power = opt->beta1_power * opt->beta1;
        opt->beta2_power = opt->beta2_power * opt->beta2;
        opt->step += 1;
    }
    crec_prof_begin(ctx, CREC_K_REDUCE_LOSS, st);
    CREC_KLAUNCH(ctx, ncf_reduce_loss, dim3((unsigned)nb), dim3(256), 0, st, (const float *)loss_inst, N, B, loss_per_batch);
    crec_prof_end(ctx, st);
#ifdef NCF_STAMPS
    {
        long long h[512];
        CREC_HIP(ctx, hipStreamSynchronize(st));
        CREC_HIP(ctx, hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost));
        for (int k = 1; k < 32 && h[2 * k]; ++k)
            fprintf(stderr, "ncf_tile stamp %2d: +%6lld clk  +%5.2f us\n", k, h[2 * k] - h[2 * k - 2], 0.01 * (double)(h[2 * k + 1] - h[2 * k - 1]));
        if (h[128])
            for (int w = 0; w < NW; ++w) {
                fprintf(stderr, "ncf_tile wave %d:", w);
                for (int j = 0; j < 8 && h[128 + w * 16 + j]; ++j)
                    fprintf(stderr, " %6lld", h[128 + w * 16 + j] - h[0]);
                fprintf(stderr, "\n");
            }
        (void)hipFree(dbg);
    }
#endif
    CREC_HIP(ctx, hipGetLastError());
    return crec_prof_flush(ctx, st);
}
