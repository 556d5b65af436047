/*
 * crec_oracle.c -- CPU restatement of the CleverRec BPR hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cleverrec_amd/ may link, import or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / the timed CPU baseline.
 *
 * Parity status
 *   - RNG + samplers + ranking metrics: PINNED.  Checked bit-for-bit against
 *     golden vectors produced by importing the reference's own
 *     utils/sampler.py and utils/metrics.py (tests/golden/, generator
 *     tools/make_golden.py) and against numpy.random.RandomState itself.
 *   - BPR graph arithmetic + Adam (TensorFlow 1.x ops; TF is a third-party
 *     dependency that is neither vendored in the reference nor installable
 *     here, unpinned ">=1.13" in reference README.md:57): PARITY UNPINNED.
 *     Restated from the graph definition model/ranking/BPR.py:31-51,
 *     utils/tools.py:66-87 and the published TF1 AdamOptimizer algorithm;
 *     cross-checked by an independent float64 autograd derivation in tests.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 * -ffp-contract=off keeps every fp32 multiply/add a separately rounded IEEE
 * operation, the way TF's Eigen CPU kernels evaluate them element-wise.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------
 * a1. NumPy legacy global RNG as the samplers use it
 *     (reference utils/sampler.py:58,60,68 -> np.random.randint / permutation;
 *      numpy RandomState = MT19937, init_genrand seeding, masked rejection).
 * --------------------------------------------------------------------- */
#define MT_N 624
#define MT_M 397

typedef struct {
    uint32_t key[MT_N];
    int32_t pos;
} orc_rng;

void orc_rng_seed(orc_rng *r, uint32_t seed)
{
    /* np.random.seed(int) == Knuth init_genrand */
    r->key[0] = seed;
    for (int k = 1; k < MT_N; k++)
        r->key[k] = 1812433253u * (r->key[k - 1] ^ (r->key[k - 1] >> 30)) + (uint32_t)k;
    r->pos = MT_N;
}

void orc_rng_set_state(orc_rng *r, const uint32_t *key, int32_t pos)
{
    memcpy(r->key, key, sizeof(r->key));
    r->pos = pos;
}

void orc_rng_get_state(const orc_rng *r, uint32_t *key, int32_t *pos)
{
    memcpy(key, r->key, sizeof(r->key));
    *pos = r->pos;
}

static void mt_refill(orc_rng *r)
{
    uint32_t *mt = r->key;
    int kk;
    uint32_t y;
    for (kk = 0; kk < MT_N - MT_M; kk++) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; kk < MT_N - 1; kk++) {
        y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
        mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    r->pos = 0;
}

uint32_t orc_next_u32(orc_rng *r)
{
    if (r->pos >= MT_N)
        mt_refill(r);
    uint32_t y = r->key[r->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* draw in [0, max] by masked rejection: one 32-bit output per trial,
 * no draw at all when max == 0 (numpy legacy rk_interval / bounded_masked). */
static uint32_t interval_u32(orc_rng *r, uint32_t max)
{
    if (max == 0)
        return 0;
    uint32_t mask = max;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    do {
        v = orc_next_u32(r) & mask;
    } while (v > max);
    return v;
}

/* np.random.randint(n), 1 <= n <= 2^32 */
int64_t orc_randint(orc_rng *r, int64_t n)
{
    return (int64_t)interval_u32(r, (uint32_t)(n - 1));
}

/* np.random.permutation(n): arange then legacy Fisher-Yates from the top */
void orc_permutation(orc_rng *r, int64_t n, int64_t *out)
{
    for (int64_t k = 0; k < n; k++)
        out[k] = k;
    for (int64_t k = n - 1; k >= 1; k--) {
        int64_t j = (int64_t)interval_u32(r, (uint32_t)k);
        int64_t t = out[k];
        out[k] = out[j];
        out[j] = t;
    }
}

/* ------------------------------------------------------------------------
 * a2-a4. Samplers (reference utils/sampler.py:10-99).
 *   ui_train is handed over as a CSR in the dict's iteration order:
 *   user_ids[k] owns items[ptr[k] .. ptr[k+1]) in list order.
 * --------------------------------------------------------------------- */
typedef struct {
    int32_t *stamp; /* stamp[item] == tag  <=>  item in seen_items */
    int32_t item_nums;
} seen_set;

static int neg_draw(orc_rng *r, const seen_set *s, int32_t tag, const int64_t *random_j, int nj, int64_t *out)
{
    /* sampler.py:58-60: j = randint(item_nums); while j in random_j or j in seen: redraw */
    for (;;) {
        int64_t j = orc_randint(r, s->item_nums);
        int bad = (s->stamp[j] == tag);
        for (int t = 0; t < nj && !bad; t++)
            bad = (random_j[t] == j);
        if (!bad) {
            *out = j;
            return 0;
        }
    }
}

static void shuffle_i64(const int64_t *perm, int64_t n, int64_t *a, int64_t *tmp, int width)
{
    /* a = np.array(a)[s_idx] */
    memcpy(tmp, a, (size_t)n * width * sizeof(int64_t));
    for (int64_t k = 0; k < n; k++)
        memcpy(a + k * width, tmp + perm[k] * width, width * sizeof(int64_t));
}

/* pairwise_ranking_sampler (utils/sampler.py:46-74). Returns N. */
int64_t orc_sample_pairwise(orc_rng *r, const int64_t *ptr, const int32_t *items, int64_t n_users,
                            const int32_t *user_ids, int32_t item_nums, int neg_ratio, int64_t *u, int64_t *i,
                            int64_t *j)
{
    seen_set s = {(int32_t *)calloc((size_t)item_nums, sizeof(int32_t)), item_nums};
    int64_t *rj = (int64_t *)malloc(sizeof(int64_t) * (size_t)(neg_ratio > 0 ? neg_ratio : 1));
    int64_t n = 0;
    for (int64_t k = 0; k < n_users; k++) {
        int32_t tag = (int32_t)(k + 1);
        for (int64_t e = ptr[k]; e < ptr[k + 1]; e++)
            s.stamp[items[e]] = tag;
        for (int64_t e = ptr[k]; e < ptr[k + 1]; e++) {
            for (int t = 0; t < neg_ratio; t++) {
                u[n] = user_ids[k];
                i[n] = items[e];
                neg_draw(r, &s, tag, rj, t, &rj[t]);
                j[n] = rj[t];
                n++;
            }
        }
    }
    int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    orc_permutation(r, n, perm); /* sampler.py:68 */
    shuffle_i64(perm, n, u, tmp, 1);
    shuffle_i64(perm, n, i, tmp, 1);
    shuffle_i64(perm, n, j, tmp, 1);
    free(perm);
    free(tmp);
    free(rj);
    free(s.stamp);
    return n;
}

/* pointwise_ranking_sampler (utils/sampler.py:10-43). Returns N. */
int64_t orc_sample_pointwise(orc_rng *r, const int64_t *ptr, const int32_t *items, int64_t n_users,
                             const int32_t *user_ids, int32_t item_nums, int neg_ratio, int64_t *u, int64_t *i,
                             double *y)
{
    seen_set s = {(int32_t *)calloc((size_t)item_nums, sizeof(int32_t)), item_nums};
    int64_t *rj = (int64_t *)malloc(sizeof(int64_t) * (size_t)(neg_ratio > 0 ? neg_ratio : 1));
    int64_t n = 0;
    for (int64_t k = 0; k < n_users; k++) {
        int32_t tag = (int32_t)(k + 1);
        for (int64_t e = ptr[k]; e < ptr[k + 1]; e++)
            s.stamp[items[e]] = tag;
        for (int64_t e = ptr[k]; e < ptr[k + 1]; e++) {
            u[n] = user_ids[k];
            i[n] = items[e];
            y[n] = 1.0;
            n++;
            for (int t = 0; t < neg_ratio; t++) {
                neg_draw(r, &s, tag, rj, t, &rj[t]);
                u[n] = user_ids[k];
                i[n] = rj[t];
                y[n] = 0.0;
                n++;
            }
        }
    }
    int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    orc_permutation(r, n, perm); /* sampler.py:37 */
    shuffle_i64(perm, n, u, tmp, 1);
    shuffle_i64(perm, n, i, tmp, 1);
    double *ytmp = (double *)tmp;
    memcpy(ytmp, y, (size_t)n * sizeof(double));
    for (int64_t k = 0; k < n; k++)
        y[k] = ytmp[perm[k]];
    free(perm);
    free(tmp);
    free(rj);
    free(s.stamp);
    return n;
}

/* ranking_sampler_cml (utils/sampler.py:77-99). neg is [N, neg_ratio]. Returns N. */
int64_t orc_sample_cml(orc_rng *r, const int64_t *ptr, const int32_t *items, int64_t n_users,
                       const int32_t *user_ids, int32_t item_nums, int neg_ratio, int64_t *u, int64_t *i,
                       int64_t *neg)
{
    seen_set s = {(int32_t *)calloc((size_t)item_nums, sizeof(int32_t)), item_nums};
    int64_t n = 0;
    for (int64_t k = 0; k < n_users; k++) {
        int32_t tag = (int32_t)(k + 1);
        for (int64_t e = ptr[k]; e < ptr[k + 1]; e++)
            s.stamp[items[e]] = tag;
        for (int64_t e = ptr[k]; e < ptr[k + 1]; e++) {
            u[n] = user_ids[k];
            i[n] = items[e];
            int64_t *row = neg + n * neg_ratio;
            for (int t = 0; t < neg_ratio; t++)
                neg_draw(r, &s, tag, row, t, &row[t]);
            n++;
        }
    }
    int64_t *perm = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t *tmp = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1) * (size_t)(neg_ratio > 0 ? neg_ratio : 1));
    orc_permutation(r, n, perm); /* sampler.py:97 */
    shuffle_i64(perm, n, u, tmp, 1);
    shuffle_i64(perm, n, i, tmp, 1);
    shuffle_i64(perm, n, neg, tmp, neg_ratio);
    free(perm);
    free(tmp);
    free(s.stamp);
    return n;
}

/* ------------------------------------------------------------------------
 * a6-a8. One BPR training step (model/ranking/BPR.py:31-44, utils/tools.py:66-87)
 *        with TF1 AdamOptimizer semantics on IndexedSlices gradients.
 *
 * TF1 log_sigmoid(x) = -softplus(-x); Eigen softplus uses the thresholded
 * form below (threshold = log(FLT_EPSILON) + 2).
 * --------------------------------------------------------------------- */
static float tf_softplus(float z)
{
    const float thr = logf(1.1920929e-07f) + 2.0f;
    if (z > -thr)
        return z;
    if (z < thr)
        return expf(z);
    return log1pf(expf(z));
}

static float dotf(const float *a, const float *b, int d)
{
    float acc = 0.0f;
    for (int k = 0; k < d; k++)
        acc = acc + a[k] * b[k];
    return acc;
}

/* lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t), powers kept as running fp32
 * products exactly like TF's beta1_power / beta2_power slot variables. */
void orc_adam_lr_table(float lr, float beta1, float beta2, int64_t first_step, int64_t n, float *b1p_io, float *b2p_io,
                       float *lr_t)
{
    float b1p = *b1p_io, b2p = *b2p_io;
    (void)first_step;
    for (int64_t k = 0; k < n; k++) {
        lr_t[k] = (lr * sqrtf(1.0f - b2p)) / (1.0f - b1p);
        b1p = b1p * beta1;
        b2p = b2p * beta2;
    }
    *b1p_io = b1p;
    *b2p_io = b2p;
}

static void adam_row(float *var, float *m, float *v, const float *g, int d, float lr_t, float beta1, float beta2,
                     float eps)
{
    const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
    for (int k = 0; k < d; k++) {
        float gk = g ? g[k] : 0.0f;
        float mk = m[k] * beta1;
        float vk = v[k] * beta2;
        if (g) {
            mk = mk + gk * omb1;
            vk = vk + (gk * gk) * omb2;
        }
        m[k] = mk;
        v[k] = vk;
        var[k] = var[k] - (lr_t * mk) / (sqrtf(vk) + eps);
    }
}

/* touched-rows-only optimizers of utils/tools.py:79-87 (TF1 semantics on IndexedSlices, duplicates summed first):
 * mode 2 GradientDescentOptimizer: var -= lr*g; mode 3 AdagradOptimizer: accum += g*g (accum kept in m, initialised
 * to 0.1 by the caller), var -= lr*g/sqrt(accum). */
static void sgd_or_adagrad_row(int mode, float *var, float *acc, const float *g, int d, float lr)
{
    for (int k = 0; k < d; k++) {
        if (mode == 2) {
            var[k] = var[k] - lr * g[k];
        } else {
            acc[k] = acc[k] + g[k] * g[k];
            var[k] = var[k] - (lr * g[k]) / sqrtf(acc[k]);
        }
    }
}

/*
 * adam_mode 2 / 3: SGD / Adagrad on the touched rows (lr_t = lr), see sgd_or_adagrad_row.
 * adam_mode 0 ("tf1"): TF1 _apply_sparse_shared: m,v decayed and var updated
 *                       on EVERY row, gradients summed per unique row first.
 * adam_mode 1 ("lazy"): only rows touched by the batch are updated.
 * Gradient rows are summed in the order TF's CPU unique+unsorted_segment_sum
 * visits them: batch position order; for Q the i-slices precede the j-slices.
 * gP/gQ: scratch [U*d], [I*d]; cntP/cntQ: scratch [U], [I] (zeroed inside).
 * loss_out (float64 accumulation of fp32 terms): sum_b softplus(-x_b)
 *   + reg*0.5*(|pu|^2+|qi|^2+|qj|^2)   (BPR.py:42-43, tools.py:70-71)
 */
void orc_bpr_step(float *P, float *Q, float *mP, float *vP, float *mQ, float *vQ, int64_t U, int64_t I, int d,
                  const int32_t *u, const int32_t *i, const int32_t *j, int64_t B, float reg, float lr_t, float beta1,
                  float beta2, float eps, int adam_mode, float *gP, float *gQ, int32_t *cntP, int32_t *cntQ,
                  double *loss_out, float *s_out)
{
    memset(gP, 0, sizeof(float) * (size_t)U * d);
    memset(gQ, 0, sizeof(float) * (size_t)I * d);
    memset(cntP, 0, sizeof(int32_t) * (size_t)U);
    memset(cntQ, 0, sizeof(int32_t) * (size_t)I);
    double loss = 0.0;
    float *sb = (float *)malloc(sizeof(float) * (size_t)(B > 0 ? B : 1));
    for (int64_t b = 0; b < B; b++) {
        const float *pu = P + (int64_t)u[b] * d, *qi = Q + (int64_t)i[b] * d, *qj = Q + (int64_t)j[b] * d;
        float x = dotf(pu, qi, d) - dotf(pu, qj, d);
        float s = 1.0f / (1.0f + expf(x)); /* sigmoid(-x) = d softplus(-x) / d(-x) */
        sb[b] = s;
        if (s_out)
            s_out[b] = s;
        float l2 = 0.5f * dotf(pu, pu, d) + 0.5f * dotf(qi, qi, d) + 0.5f * dotf(qj, qj, d);
        loss += (double)tf_softplus(-x) + (double)(reg * l2);
        float *g = gP + (int64_t)u[b] * d;
        for (int k = 0; k < d; k++)
            g[k] = g[k] + ((-s) * (qi[k] - qj[k]) + reg * pu[k]);
        cntP[u[b]]++;
    }
    for (int64_t b = 0; b < B; b++) { /* i-slices */
        const float *pu = P + (int64_t)u[b] * d, *qi = Q + (int64_t)i[b] * d;
        float *g = gQ + (int64_t)i[b] * d;
        for (int k = 0; k < d; k++)
            g[k] = g[k] + ((-sb[b]) * pu[k] + reg * qi[k]);
        cntQ[i[b]]++;
    }
    for (int64_t b = 0; b < B; b++) { /* j-slices */
        const float *pu = P + (int64_t)u[b] * d, *qj = Q + (int64_t)j[b] * d;
        float *g = gQ + (int64_t)j[b] * d;
        for (int k = 0; k < d; k++)
            g[k] = g[k] + (sb[b] * pu[k] + reg * qj[k]);
        cntQ[j[b]]++;
    }
    for (int64_t r = 0; r < U; r++) {
        if (cntP[r] && adam_mode >= 2)
            sgd_or_adagrad_row(adam_mode, P + r * d, mP + r * d, gP + r * d, d, lr_t);
        else if (cntP[r])
            adam_row(P + r * d, mP + r * d, vP + r * d, gP + r * d, d, lr_t, beta1, beta2, eps);
        else if (adam_mode == 0)
            adam_row(P + r * d, mP + r * d, vP + r * d, NULL, d, lr_t, beta1, beta2, eps);
    }
    for (int64_t r = 0; r < I; r++) {
        if (cntQ[r] && adam_mode >= 2)
            sgd_or_adagrad_row(adam_mode, Q + r * d, mQ + r * d, gQ + r * d, d, lr_t);
        else if (cntQ[r])
            adam_row(Q + r * d, mQ + r * d, vQ + r * d, gQ + r * d, d, lr_t, beta1, beta2, eps);
        else if (adam_mode == 0)
            adam_row(Q + r * d, mQ + r * d, vQ + r * d, NULL, d, lr_t, beta1, beta2, eps);
    }
    free(sb);
    *loss_out = loss;
}

/* ------------------------------------------------------------------------
 * a13. Full-catalogue scores + seen filter + top-K
 *      (model/RankingRecommender.py:198-247, model/ranking/BPR.py:51).
 * order == 0: plain k = 0..d-1 fp32 chain.
 * order == 1: the k order of the gfx950 v_mfma_f32_32x32x2_f32 chain the HIP
 *             kernel issues (step s uses k = s and k = DMAX/2 + s, in that order),
 *             each product folded in with one fmaf rounding.
 * Ranking: descending score (ascending when `ascending`), ties -> lower item id
 * (the reference's np.argsort is not stable, so its tie order is unspecified).
 * Users whose seen list is empty skip the filter (RankingRecommender.py:231-232).
 * Fewer than K unseen items: remaining slots are -1 (reference would index out
 * of range).
 * --------------------------------------------------------------------- */
void orc_scores(const float *P, const float *Q, int64_t I, int d, const int32_t *users, int64_t Bt, int order,
                float *scores)
{
    for (int64_t r = 0; r < Bt; r++) {
        const float *pu = P + (int64_t)users[r] * d;
        for (int64_t c = 0; c < I; c++) {
            const float *q = Q + c * d;
            float acc = 0.0f;
            if (order == 0) {
                for (int k = 0; k < d; k++)
                    acc = acc + pu[k] * q[k];
            } else {
                /* the kernel pads d up to DMAX in {32, 64, 128, 256}: lanes 0-31 take k = s,
                 * lanes 32-63 take k = DMAX/2 + s; products with k >= d are 0*0 and leave the
                 * chain unchanged, so they are skipped here */
                int dmax = d <= 32 ? 32 : d <= 64 ? 64 : d <= 128 ? 128 : 256;
                int h = dmax / 2;
                for (int s = 0; s < h; s++) {
                    if (s < d)
                        acc = fmaf(pu[s], q[s], acc);
                    if (h + s < d)
                        acc = fmaf(pu[h + s], q[h + s], acc);
                }
            }
            scores[r * I + c] = acc;
        }
    }
}

static int better(float sa, int32_t ia, float sb, int32_t ib, int ascending)
{
    if (sa != sb)
        return ascending ? (sa < sb) : (sa > sb);
    return ia < ib;
}

void orc_topk(const float *scores, int64_t I, const int32_t *users, int64_t Bt, const int64_t *seen_ptr,
              const int32_t *seen_items, int K, int ascending, int32_t *topk_ids, float *topk_scores)
{
    uint8_t *seen = (uint8_t *)calloc((size_t)I, 1);
    for (int64_t r = 0; r < Bt; r++) {
        int32_t uu = users[r];
        for (int64_t e = seen_ptr[uu]; e < seen_ptr[uu + 1]; e++)
            seen[seen_items[e]] = 1;
        int32_t *ids = topk_ids + r * K;
        float *sc = topk_scores + r * K;
        int n = 0;
        const float *row = scores + r * I;
        for (int64_t c = 0; c < I; c++) {
            if (seen[c])
                continue;
            float s = row[c];
            if (n == K && !better(s, (int32_t)c, sc[K - 1], ids[K - 1], ascending))
                continue;
            int pos = (n < K) ? n : K - 1;
            while (pos > 0 && better(s, (int32_t)c, sc[pos - 1], ids[pos - 1], ascending)) {
                sc[pos] = sc[pos - 1];
                ids[pos] = ids[pos - 1];
                pos--;
            }
            sc[pos] = s;
            ids[pos] = (int32_t)c;
            if (n < K)
                n++;
        }
        for (; n < K; n++) {
            ids[n] = -1;
            sc[n] = ascending ? INFINITY : -INFINITY;
        }
        for (int64_t e = seen_ptr[uu]; e < seen_ptr[uu + 1]; e++)
            seen[seen_items[e]] = 0;
    }
    free(seen);
}

/* ------------------------------------------------------------------------
 * a14. cal_ranking_metrics (utils/metrics.py:9-19), float64.
 *  hit / min(K, len(real)), sum of reciprocal ranks of ALL hits,
 *  dcg / idcg with idcg taken over ALL true items (not capped at K).
 *  rec = topk_ids[r, :K'] for each K' in ks; test CSR indexed by user id.
 * --------------------------------------------------------------------- */
void orc_rank_metrics(const int32_t *topk_ids, const int32_t *users, int64_t Bt, int Kmax, const int32_t *ks, int nk,
                      const int64_t *test_ptr, const int32_t *test_items, double *hr, double *mrr, double *ndcg)
{
    for (int q = 0; q < nk; q++) {
        int K = ks[q];
        for (int64_t r = 0; r < Bt; r++) {
            const int32_t *rec = topk_ids + r * Kmax;
            int32_t uu = users[r];
            int64_t n_real = test_ptr[uu + 1] - test_ptr[uu];
            double hit = 0, rr = 0, dcg = 0, idcg = 0;
            for (int64_t id = 0; id < n_real; id++) {
                int32_t item = test_items[test_ptr[uu] + id];
                for (int idx = 0; idx < K; idx++) {
                    if (rec[idx] == item) { /* np.where(rec == item)[0][0]: first match */
                        hit += 1;
                        rr += 1.0 / (idx + 1);
                        dcg += 1.0 / log2((double)(idx + 2));
                        break;
                    }
                }
                idcg += 1.0 / log2((double)(id + 2));
            }
            double denom = (double)(K < n_real ? K : n_real);
            hr[q * Bt + r] = hit / denom;
            mrr[q * Bt + r] = rr;
            ndcg[q * Bt + r] = dcg / idcg;
        }
    }
}

/* ------------------------------------------------------------------------
 * Data-parallel form of the same step: users sharded over ranks, Q replicated.
 * orc_bpr_dp_local: this rank's triples -> tf1 Adam on its P shard, dense partial dQ [I,d]
 *                   (i-slices then j-slices, position order), local loss.
 * orc_adam_dense  : tf1 Adam of every Q row from the rank-summed gradient (a row nobody touched has
 *                   g == 0, which is exactly the untouched sweep).
 * Summing dQ over ranks and applying orc_adam_dense on every replica equals orc_bpr_step on the
 * concatenated batch up to fp32 add order.
 * --------------------------------------------------------------------- */
void orc_bpr_dp_local(float *P, const float *Q, float *mP, float *vP, int64_t U, int64_t I, int d, const int32_t *u,
                      const int32_t *i, const int32_t *j, int64_t B, float reg, float lr_t, float beta1, float beta2,
                      float eps, float *gP, int32_t *cntP, float *dQ, double *loss_out)
{
    memset(gP, 0, sizeof(float) * (size_t)U * d);
    memset(cntP, 0, sizeof(int32_t) * (size_t)U);
    memset(dQ, 0, sizeof(float) * (size_t)I * d);
    double loss = 0.0;
    float *sb = (float *)malloc(sizeof(float) * (size_t)(B > 0 ? B : 1));
    for (int64_t b = 0; b < B; b++) {
        const float *pu = P + (int64_t)u[b] * d, *qi = Q + (int64_t)i[b] * d, *qj = Q + (int64_t)j[b] * d;
        float x = dotf(pu, qi, d) - dotf(pu, qj, d);
        float s = 1.0f / (1.0f + expf(x));
        sb[b] = s;
        float l2 = 0.5f * dotf(pu, pu, d) + 0.5f * dotf(qi, qi, d) + 0.5f * dotf(qj, qj, d);
        loss += (double)tf_softplus(-x) + (double)(reg * l2);
        float *g = gP + (int64_t)u[b] * d;
        for (int k = 0; k < d; k++)
            g[k] = g[k] + ((-s) * (qi[k] - qj[k]) + reg * pu[k]);
        cntP[u[b]]++;
    }
    for (int64_t b = 0; b < B; b++) {
        const float *pu = P + (int64_t)u[b] * d, *qi = Q + (int64_t)i[b] * d;
        float *g = dQ + (int64_t)i[b] * d;
        for (int k = 0; k < d; k++)
            g[k] = g[k] + ((-sb[b]) * pu[k] + reg * qi[k]);
    }
    for (int64_t b = 0; b < B; b++) {
        const float *pu = P + (int64_t)u[b] * d, *qj = Q + (int64_t)j[b] * d;
        float *g = dQ + (int64_t)j[b] * d;
        for (int k = 0; k < d; k++)
            g[k] = g[k] + (sb[b] * pu[k] + reg * qj[k]);
    }
    for (int64_t r = 0; r < U; r++)
        adam_row(P + r * d, mP + r * d, vP + r * d, cntP[r] ? gP + r * d : NULL, d, lr_t, beta1, beta2, eps);
    free(sb);
    *loss_out = loss;
}

void orc_adam_dense(float *var, float *m, float *v, const float *g, int64_t rows, int d, float lr_t, float beta1,
                    float beta2, float eps)
{
    for (int64_t r = 0; r < rows; r++)
        adam_row(var + r * d, m + r * d, v + r * d, g + r * d, d, lr_t, beta1, beta2, eps);
}

/* ------------------------------------------------------------------------
 * Row-sharded form of the lazy step (include/crec.h "row-sharded training"): users sharded over
 * ranks, items owned round-robin.  The arithmetic of one rank, on a full-size Q for simplicity (the oracle does
 * not model where rows live, only in which order sums are formed):
 * orc_bpr_shard_local: this rank's triples -> lazy Adam on the P rows they touch; for every item row they
 *                   reference, the rank's gradient sum dQ (i-slices then j-slices, position order, l2 term
 *                   per occurrence) and the occurrence count cntQ; local loss.
 * orc_adam_rows_lazy : lazy Adam of the rows with touched[r] != 0 from a gradient the caller has summed over
 *                   ranks in rank order.
 * Summing dQ over ranks and applying orc_adam_rows_lazy equals orc_bpr_step(adam_mode 1) on the concatenated
 * batch up to fp32 add order.
 * --------------------------------------------------------------------- */
void orc_bpr_shard_local(float *P, const float *Q, float *mP, float *vP, int64_t U, int64_t I, int d, const int32_t *u,
                         const int32_t *i, const int32_t *j, int64_t B, float reg, float lr_t, float beta1, float beta2,
                         float eps, float *gP, int32_t *cntP, float *dQ, int32_t *cntQ, double *loss_out)
{
    memset(gP, 0, sizeof(float) * (size_t)U * d);
    memset(cntP, 0, sizeof(int32_t) * (size_t)U);
    memset(dQ, 0, sizeof(float) * (size_t)I * d);
    memset(cntQ, 0, sizeof(int32_t) * (size_t)I);
    double loss = 0.0;
    float *sb = (float *)malloc(sizeof(float) * (size_t)(B > 0 ? B : 1));
    for (int64_t b = 0; b < B; b++) {
        const float *pu = P + (int64_t)u[b] * d, *qi = Q + (int64_t)i[b] * d, *qj = Q + (int64_t)j[b] * d;
        float x = dotf(pu, qi, d) - dotf(pu, qj, d);
        float s = 1.0f / (1.0f + expf(x));
        sb[b] = s;
        float l2 = 0.5f * dotf(pu, pu, d) + 0.5f * dotf(qi, qi, d) + 0.5f * dotf(qj, qj, d);
        loss += (double)tf_softplus(-x) + (double)(reg * l2);
        float *g = gP + (int64_t)u[b] * d;
        for (int k = 0; k < d; k++)
            g[k] = g[k] + ((-s) * (qi[k] - qj[k]) + reg * pu[k]);
        cntP[u[b]]++;
    }
    for (int64_t b = 0; b < B; b++) {
        const float *pu = P + (int64_t)u[b] * d, *qi = Q + (int64_t)i[b] * d;
        float *g = dQ + (int64_t)i[b] * d;
        for (int k = 0; k < d; k++)
            g[k] = g[k] + ((-sb[b]) * pu[k] + reg * qi[k]);
        cntQ[i[b]]++;
    }
    for (int64_t b = 0; b < B; b++) {
        const float *pu = P + (int64_t)u[b] * d, *qj = Q + (int64_t)j[b] * d;
        float *g = dQ + (int64_t)j[b] * d;
        for (int k = 0; k < d; k++)
            g[k] = g[k] + (sb[b] * pu[k] + reg * qj[k]);
        cntQ[j[b]]++;
    }
    for (int64_t r = 0; r < U; r++)
        if (cntP[r])
            adam_row(P + r * d, mP + r * d, vP + r * d, gP + r * d, d, lr_t, beta1, beta2, eps);
    free(sb);
    *loss_out = loss;
}

void orc_adam_rows_lazy(float *var, float *m, float *v, const float *g, const int32_t *touched, int64_t rows, int d,
                        float lr_t, float beta1, float beta2, float eps)
{
    for (int64_t r = 0; r < rows; r++)
        if (touched[r])
            adam_row(var + r * d, m + r * d, v + r * d, g + r * d, d, lr_t, beta1, beta2, eps);
}
