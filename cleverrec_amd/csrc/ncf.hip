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
//                   gradients of the small dense parameters (W_k, b_k, h).
//   crec_apply_all  (rows_adam.hip) every parameter update in one launch: per embedding table, contribution rows
//                   summed per table row in batch order (+ reg * row per occurrence), TF1 sparse Adam over every
//                   row; the small parameters: tile partials added in 16 fixed slices, TF1 dense ApplyAdam.
// Everything is deterministic (fixed summation orders, no atomics).
#include "crec_internal.h"
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
__global__ __launch_bounds__(NT) void ncf_tile(NcfDev m, const int32_t *__restrict__ u, const int32_t *__restrict__ i,
                                               const float *__restrict__ y, int nb, float *__restrict__ c_pg,
                                               float *__restrict__ c_qg, float *__restrict__ c_pm,
                                               float *__restrict__ c_qm, float *__restrict__ loss_inst,
                                               float *__restrict__ partial)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
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
    const bool has_g = m.kind != 1, has_m = m.kind != 0;
    const int dg = has_g ? m.dg : 0, L = has_m ? m.n_layers : 0, hm = m.L0 / 2, ldg = dg + 1;

    // carve LDS (ncf_tile_lds_floats below mirrors this).  Widths halve exactly (ncf_describe), so the offsets
    // are closed-form scalar arithmetic: no pointer tables (they would live in scratch)
    const auto width_of = [&](int k) { return has_m ? (m.L0 >> k) : 0; };
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
    int *const su = (int *)(p + 2 * T), *const si = su + T;
    float *const b_base = (float *)(si + T);
    float *const hl = b_base + boff_of(L); // GMF: h_gmf[dg]; MLP: h_mlp[last]; NeuMF: h_neumf[dg + last]
    float *const W_base = hl + dg + last;
#define WL_(k) (W_base + woff_of(k))
#define BL_(k) (b_base + boff_of(k))

    if (tid < T) {
        const bool ok = tid < nr;
        su[tid] = ok ? u[r0 + tid] : 0;
        si[tid] = ok ? i[r0 + tid] : 0;
        yl[tid] = ok ? y[r0 + tid] : 0.f;
    }
    // every parameter on chip for both passes; these loads overlap the index fetch
    for (int c = tid; c < dg + last; c += NT)
        hl[c] = m.small[m.h_off + c];
    for (int k = 0; k < L; ++k) {
        for (int c = tid; c < W_(k + 1); c += NT)
            BL_(k)[c] = m.small[m.b_off[k] + c];
        stage_weights<NT>(WL_(k), m.small + m.w_off[k], W_(k), W_(k + 1));
    }
    __syncthreads();
    STAMP(); // 1: parameters staged
    // ---- gathers: a wave owns RW rows, lanes run along the row; all RW rows' loads in flight together ------------
    {
        int64_t ub[RW], ib[RW];
        bool ok[RW];
#pragma unroll
        for (int rr = 0; rr < RW; ++rr) {
            const int r = wave * RW + rr;
            ok[rr] = r < nr;
            ub[rr] = su[r];
            ib[rr] = si[r];
        }
        for (int c = lane; c < dg; c += 64) {
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
            for (int c = lane; c < hm; c += 64) {
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
    for (int k = 0; k < L; ++k) {
        const int kin = W_(k), kout = W_(k + 1), ws = kout + 1, lda = LD_(k), ldo = LD_(k + 1);
        const float *b = BL_(k), *Ak = A_(k), *Wl = WL_(k);
        float *An = A_(k + 1);
        __syncthreads(); // A[k] complete
        tile_gemm<NW>(
            T, kout, kin, [&](int r, int kk) { return Ak[r * lda + kk]; }, [&](int kk, int c) { return Wl[kk * ws + c]; },
            [&](int r, int c, float s) {
                s += b[c];
                An[r * ldo + c] = s > 0.f ? s : 0.f;
            });
    }
    __syncthreads();
    STAMP(); // 3..: forward
    // ---- logit, e, loss: 8 lanes per instance, each over a contiguous eighth of the columns, then a fixed
    // butterfly over the 8 -----------------------------------------------------------------------------------------
    if (tid < T * 8) {
        const int r = tid >> 3, sub = tid & 7;
        const auto eighth = [&](int n, int &c0, int &c1) {
            const int cw = (n + 7) >> 3;
            c0 = sub * cw < n ? sub * cw : n;
            c1 = c0 + cw < n ? c0 + cw : n;
        };
        float zg = 0.f, ng = 0.f, zm = 0.f, nm = 0.f;
        int c0, c1;
        eighth(dg, c0, c1);
        for (int c = c0; c < c1; ++c) {
            const float pg = G_p[r * ldg + c], qg = G_q[r * ldg + c];
            zg = zg + (pg * qg) * hl[c];
            ng = ng + (pg * pg + qg * qg);
        }
        if (has_m) {
            const float *AL = A_(L), *A0 = A_(0);
            const int ldl = LD_(L), ld0 = LD_(0);
            eighth(last, c0, c1);
            for (int c = c0; c < c1; ++c)
                zm = zm + AL[r * ldl + c] * hl[dg + c];
            eighth(W_(0), c0, c1);
            for (int c = c0; c < c1; ++c) {
                const float a = A0[r * ld0 + c];
                nm = nm + a * a;
            }
        }
#pragma unroll
        for (int off = 1; off < 8; off <<= 1) {
            zg += __shfl_xor(zg, off, 64);
            ng += __shfl_xor(ng, off, 64);
            zm += __shfl_xor(zm, off, 64);
            nm += __shfl_xor(nm, off, 64);
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
    float *part = partial + (int64_t)tile * m.n_small;
    STAMP(); // logit
    // ---- GMF backward ----------------------------------------------------------------------------------------
    if (has_g) {
        for (int c = lane; c < dg; c += 64) {
            const float hc = hl[c];
#pragma unroll
            for (int rr = 0; rr < RW; ++rr) {
                const int r = wave * RW + rr;
                if (r < nr) {
                    const float er = e[r];
                    c_pg[(int64_t)(r0 + r) * dg + c] = er * (G_q[r * ldg + c] * hc);
                    c_qg[(int64_t)(r0 + r) * dg + c] = er * (G_p[r * ldg + c] * hc);
                }
            }
        }
        for (int c = tid; c < dg; c += NT) { // dL/dh_g (rows past nr carry e = 0)
            float s = 0.f;
#pragma unroll 16
            for (int r = 0; r < T; ++r)
                s = s + e[r] * (G_p[r * ldg + c] * G_q[r * ldg + c]);
            part[m.h_off + c] = s;
        }
    }
    if (has_m) {
        const float *AL = A_(L);
        float *DL = D_(L);
        const int ldl = LD_(L);
        for (int c = tid; c < last; c += NT) { // dL/dh_m
            float s = 0.f;
#pragma unroll 16
            for (int r = 0; r < T; ++r)
                s = s + e[r] * AL[r * ldl + c];
            part[m.h_off + dg + c] = s;
        }
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
        for (int k = L - 1; k >= 0; --k) {
            const int kin = W_(k), kout = W_(k + 1), ws = kout + 1, lda = LD_(k), ldd = LD_(k + 1);
            const float *Ak = A_(k), *Dn = D_(k + 1), *Wl = WL_(k);
            float *Dk = D_(k);
            float *pw = part + m.w_off[k];
            __syncthreads(); // D[k+1] complete
            // dW_k[kk][c] = sum_r A[k][r][kk] * D[k+1][r][c]   (rows past nr carry D = 0)
            tile_gemm<NW>(
                    kin, kout, T, [&](int kk, int r) { return Ak[r * lda + kk]; }, [&](int r, int c) { return Dn[r * ldd + c]; },
                    [&](int kk, int c, float s) { pw[kk * kout + c] = s; });
            for (int c = tid; c < kout; c += NT) { // db_k[c] = sum_r D[k+1][r][c]
                float s = 0.f;
#pragma unroll 16
                for (int r = 0; r < T; ++r)
                    s = s + Dn[r * ldd + c];
                part[m.b_off[k] + c] = s;
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
            else
                tile_gemm<NW>(
                    T, kin, kout, [&](int r, int c) { return Dn[r * ldd + c]; }, [&](int c, int kk) { return Wl[kk * ws + c]; },
                    [&](int r, int kk, float s) {
                        if (r < nr) {
                            if (kk < hm)
                                c_pm[(int64_t)(r0 + r) * hm + kk] = s;
                            else
                                c_qm[(int64_t)(r0 + r) * hm + (kk - hm)] = s;
                        }
                    });
            STAMP(); // dX_k
        }
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
    CREC_REQUIRE(ctx, opt->adam_mode == CREC_ADAM_TF1, "the NCF step implements adam_mode tf1");
    CREC_REQUIRE(ctx, B >= 1 && B <= 16384, "batch_size %d: at most 16384", B);
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
    const size_t ixu = crec_rows_index_bytes(nb, B, U), ixi = crec_rows_index_bytes(nb, B, I);
    const size_t cb = sizeof(float) * (size_t)B * (2 * d.dg + 2 * hm);
    const size_t pb = sizeof(float) * (size_t)n_tiles * d.n_small, lb = sizeof(float) * (size_t)nb * B;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    if (int rc = crec_reserve(ctx, &ctx->idx_ws, up(ixu) + up(ixi) + up(cb) + up(pb) + up(lb)))
        return rc;
    char *ws = (char *)ctx->idx_ws.ptr;
    crec_rows_index_view vu, vi;
    if (int rc = crec_rows_index(ctx, u, N, B, U, ws, &vu, st))
        return rc;
    if (int rc = crec_rows_index(ctx, i, N, B, I, ws + up(ixu), &vi, st))
        return rc;
    float *c_pg = (float *)(ws + up(ixu) + up(ixi));
    float *c_qg = c_pg + (size_t)B * d.dg, *c_pm = c_qg + (size_t)B * d.dg, *c_qm = c_pm + (size_t)B * hm;
    float *partial = (float *)(ws + up(ixu) + up(ixi) + up(cb));
    float *loss_inst = (float *)(ws + up(ixu) + up(ixi) + up(cb) + up(pb));

    // LDS of the tile kernel
    const size_t lds_f = ncf_tile_lds_floats(has_g, has_m, d.dg, d.L0, d.n_layers);
    const size_t lds = lds_f * sizeof(float);
    CREC_REQUIRE(ctx, lds <= 160 * 1024, "model too wide for the on-chip tile (%zu bytes of LDS)", lds);
    CREC_HIP(ctx, hipFuncSetAttribute((const void *)ncf_tile, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

#ifdef NCF_STAMPS
    long long *dbg = nullptr;
    CREC_HIP(ctx, hipMalloc(&dbg, 64 * sizeof(long long)));
    CREC_HIP(ctx, hipMemset(dbg, 0, 64 * sizeof(long long)));
    d.dbg = dbg;
#endif
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t t0 = b * B;
        const int nbt = (int)((N - t0) < B ? (N - t0) : B);
        crec_rows_opt o{};
        o.kind = ROWS_SPARSE_TF1;
        o.lr_t = (opt->lr * sqrtf(1.0f - opt->beta2_power)) / (1.0f - opt->beta1_power);
        o.beta1 = opt->beta1;
        o.beta2 = opt->beta2;
        o.eps = opt->eps;
        crec_prof_begin(ctx, CREC_K_NCF_TILE, st);
        CREC_KLAUNCH(ctx, ncf_tile, dim3((unsigned)crec_ceil_div(nbt, T)), dim3(NT), lds, st, d, u + t0, i + t0, y + t0, nbt,
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
        long long h[64];
        CREC_HIP(ctx, hipStreamSynchronize(st));
        CREC_HIP(ctx, hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost));
        for (int k = 1; k < 32 && h[2 * k]; ++k)
            fprintf(stderr, "ncf_tile stamp %2d: +%6lld clk  +%5.2f us\n", k, h[2 * k] - h[2 * k - 2], 0.01 * (double)(h[2 * k + 1] - h[2 * k - 1]));
        (void)hipFree(dbg);
    }
#endif
    CREC_HIP(ctx, hipGetLastError());
    return crec_prof_flush(ctx, st);
}
