/*
 * crec.h -- C ABI of libcrec.so, the MI355X (gfx950) embedding-training core that
 * sits under CleverRec's Recommender / RankingRecommender classes.
 *
 * The reference has no FFI of its own: its "operator API" for this path is the
 * per-batch `sess.run` into a TF1 graph plus three pure-Python helpers.  Each entry
 * point below names the reference interface it replaces (paths relative to the
 * reference root); INTEGRATION.md shows the ctypes stub a reference maintainer adds.
 *
 * Conventions
 *   - every function returns CREC_OK (0) or a negative CREC_E* code and never throws;
 *     crec_last_error(ctx) holds the text of the last failure on that ctx.
 *   - the caller owns every buffer it passes; the library owns only the opaque ctx
 *     (index workspace, contribution rows, graphs).  No ownership is transferred.
 *   - pointers marked [dev] are device (HBM) addresses on the ctx's GPU, [host] are
 *     host addresses.  `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     device work is enqueued on it and not synchronised unless stated.
 *   - one ctx per (device, host thread); a ctx is not thread-safe, distinct ctxs are
 *     independent.
 *   - embedding tables are dense row-major fp32 [rows, d]; d % 4 == 0, 4 <= d <= 256;
 *     table base addresses 16-byte aligned.
 */
#ifndef CREC_H
#define CREC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CREC_OK 0
#define CREC_EINVAL (-1)       /* bad argument (shape, alignment, null pointer, range) */
#define CREC_EHIP (-2)         /* a HIP runtime call or kernel launch failed */
#define CREC_ENOMEM (-3)       /* host or device allocation failed */
#define CREC_EUNSUPPORTED (-4) /* valid request this build does not implement */
#define CREC_ENODEVICE (-5)    /* no usable gfx950 device */

#define CREC_ADAM_TF1 0  /* TF1 AdamOptimizer on IndexedSlices: every row decays and moves each step */
#define CREC_ADAM_LAZY 1 /* only rows touched by the batch are updated */
/* the other two optimizers of utils/tools.py:79-87, in the same field (crec_adam.adam_mode); both update the rows a batch
 * touches only, which IS their TF1 semantics on IndexedSlices (gradients of duplicate indices summed first):
 *   SGD      tf.train.GradientDescentOptimizer(lr): var -= lr * g
 *   Adagrad  tf.train.AdagradOptimizer(lr): accum += g*g; var -= lr * g / sqrt(accum); the accumulator is the table's `m`
 *            slot, which the caller initialises to 0.1 (initial_accumulator_value); `v` is unused */
#define CREC_OPT_SGD 2
#define CREC_OPT_ADAGRAD 3
/* TF1-exact Adam with the traffic of the lazy step, for tables too large to sweep every step (BASELINE configs[4]:
 * a TF1 sweep of 10 M x 1 M x 128 is 16.9 GB per batch).  TF1 decays m, v and moves var on EVERY row at every step
 * (SURVEY a7); here a row carries the step at which its state was last brought up to date and replays the steps it sat
 * out whenever a batch touches it -- before its values are used -- and, for all rows, in crec_bpr_adam_flush.
 * Only through crec_bpr_train_epoch_deferred. */
#define CREC_ADAM_DEFERRED 4

typedef struct crec_ctx crec_ctx;

int crec_version(void);

/* device >= 0: bind to that HIP device (replaces tf.Session + CUDA_VISIBLE_DEVICES=gpu.id,
 * main.py:40-47).  device < 0: host-only ctx (RNG + samplers), usable without a GPU. */
int crec_ctx_create(int device, crec_ctx **out);
void crec_ctx_destroy(crec_ctx *ctx);
/* ctx == NULL: the reason the last crec_ctx_create on this process failed */
const char *crec_last_error(const crec_ctx *ctx);
/* Tuning / test options of a ctx (no reference counterpart: TF1 has no such knobs).  Defaults come from the
 * environment ONCE at crec_ctx_create (CREC_CH, CREC_WIDE, CREC_WIDE_ROUND, CREC_EVAL_SHAPE, CREC_EVAL_NO_BITMAP);
 * results never depend on them beyond fp32 summation order ("ch") -- they move work between waves.
 *   "ch"            occurrences per heavy-row chunk of the BPR step (0 = default: 64 tf1, 16 lazy; 1..255)
 *   "wide"          tf1 step: occurrences per row wave above which a whole workgroup takes it (-1 default 16, 0 off)
 *   "wide_round"    experiment: pad the wide groups' workgroups to a multiple of this (0 = the chunk's maximum)
 *   "eval_shape"    4 / 8: force an eval_sweep workgroup shape where legal (0 = choose)
 *   "eval_no_bitmap" 1: seen filter by binary search even where per-user bitmaps would fit
 *   "ncf_generic"    1: the generic NCF / LRML tile kernels even for shapes that have a shape-specialised one
 *   "no_zero_copy"   1: the tf1 step's per-chunk counts come back by copy + stream synchronisation instead of being written
 *                    into page-locked memory by the index kernel itself
 *   "idx_one_wg"     1: the tf1 step's per-chunk index with one workgroup per (batch, side) instead of two that share the phases
 *                    behind the sort (same index, bit for bit)
 *   "dp_unfused"     1: the replicated-Q step over the peer transport as an all-reduce of dQ plus a separate Adam launch instead of
 *                    one launch that exchanges and updates (same tables, bit for bit)
 *   "flow"           tf1 step (replaces the per-batch sess.run of model/RankingRecommender.py:39-47): -1 / 1 (default) all steps of a
 *                    chunk of batches in ONE dataflow launch (bpr_steps_flow: a lane group owns a table row for up to 64 steps, the
 *                    row and its Adam slots stay in registers, a version of every row is published per step) where every wave of
 *                    it is resident at once (tables of up to ~14 000 rows at d = 64), 0 a launch per step (bpr_step_fused: any
 *                    table, the data-parallel step).  The two add a row's contributions in different orders: same results up to
 *                    fp32 summation order
 *   "flow_steps"     steps per bpr_steps_flow launch (0 = up to 64; 1 = a launch per step through the same kernel: bit-identical)
 *   "flow_deal"      0: its rows are dealt to workgroups round-robin by id instead of by how often they occur (bit-identical)
 *   "lazy_window"    touched-rows steps (lazy / deferred Adam, SGD, Adagrad; LDS-indexed batches, not the row-sharded step): W = 2..16 consecutive batches of
 *                    bpr_triples_lazy per launch, a row's readers waiting for the batch before them that wrote it (bit-identical to
 *                    0 = a launch per batch).  -1 (default; env CREC_LAZY_WINDOW): 12 where both tables hold >= 3 batches' worth of
 *                    rows and d >= 32 (S-big: 18.6 -> 13.4 us/step, uniform positives 21.4 -> 13.7), else 0 (small catalogues, where
 *                    every row recurs from batch to batch and the launch is bound by the rows' way through it: ml-1m +13 %)
 *   "tf1_sweep"      1: adam_mode tf1 sweeps every row at every step whatever the tables' size (see crec_bpr_train_epoch)
 *   "no_call_tail"   1: a tf1 call that ends in the ctx's second buffer set copies its tables home in a launch of its own (default 0:
 *                    the copy rides in the last chunk's loss reduction launch -- a launch boundary less per call; the same tables)
 *   "idx_no_ids16"   1: the tf1 index gathers its occurrence records' ids from the streams instead of from the 16-bit copies it keeps in LDS
 *                    where both catalogues are below 65 536 rows (the same records)
 *   "flow_abl"       timing ablations of bpr_steps_flow / the window (results are WRONG; tools/ only)
 *   "big_index"      1: the lazy-family steps index their batches through the global-memory sort (index_big.hip, the form batches of
 *                    more than 8192 triples always take) even where the in-LDS sort would do: same index, bit for bit
 *   "idx_overlap"    1: that form indexes its chunks of batches a chunk ahead on a side stream, beside the previous chunk's steps
 *                    (two index buffers), instead of in line on the caller's stream: same results.  Off by default: on one GPU
 *                    the step kernel slows down by what the index kernels take (measured: 135 against 133 us per step) */
int crec_ctx_set_option(crec_ctx *ctx, const char *name, int64_t value);

/* ---- measurement: per-kernel time from HIP events recorded on the launch stream ------------
 * While enabled, every kernel launch of this ctx is bracketed by two events and each API call
 * ends with a stream synchronise (so do not time whole-job throughput with it on).
 * kernel_class: 0 bpr_index_batches, 1 bpr_step_fused, 2 bpr_triples_lazy, 3 bpr_steps_flow (a launch holds up to 64 steps:
 *               crec_profile_read_units gives the steps of the counted launches),
 *               4 bpr_reduce_loss, 5 eval_sweep, 6 eval_merge, 7 rank_metrics, 8 (unused), 9 bpr_adam_dense,
 *               10 rows_index_batches, 11 rows_apply, 12 small_adam_dense, 13 ncf_tile, 14 fm_instances, 15 cml_instances,
 *               16 ncf_pair_logits, 17 shard_plan_batches, 18 shard_serve_rows, 19 shard_owner_apply,
 *               20 peer_allreduce_adam.
 * enable(.., 1) also clears the accumulated totals. */
int crec_profile_enable(crec_ctx *ctx, int on);
int crec_profile_read(const crec_ctx *ctx, int kernel_class, double *total_ms, int64_t *launches);
int crec_profile_read_units(const crec_ctx *ctx, int kernel_class, int64_t *units);

/* ---- RNG: the NumPy legacy global MT19937 stream the samplers draw from -------------
 * replaces np.random.seed / get_state / set_state as used around utils/sampler.py. */
int crec_rng_seed(crec_ctx *ctx, uint32_t seed);
int crec_rng_set_state(crec_ctx *ctx, const uint32_t key[624], int32_t pos);
int crec_rng_get_state(const crec_ctx *ctx, uint32_t key[624], int32_t *pos);
/* np.random.randint(n), 1 <= n <= 2^32 (exposed for tests and for callers that interleave draws) */
int crec_rng_randint(crec_ctx *ctx, int64_t n, int64_t *out);
/* np.random.permutation(n) -> out[n] [host] */
int crec_rng_permutation(crec_ctx *ctx, int64_t n, int64_t *out);

/* ---- samplers (host, bit-exact with the reference on the same RNG state) -------------
 * ui_train is a CSR in the dict's iteration order: user_ids[k] owns
 * items[ptr[k] .. ptr[k+1]) in list order.  Outputs [host], already shuffled by the
 * epoch permutation; *n_out = number of instances written.
 *   pairwise : utils/sampler.py:46-74   capacity n_pos*neg_ratio      each of u, i, j
 *   pointwise: utils/sampler.py:10-43   capacity n_pos*(1+neg_ratio)  each of u, i, y
 *   cml      : utils/sampler.py:77-99   capacity n_pos for u, i; n_pos*neg_ratio for neg
 * Fails with CREC_EINVAL when some user has fewer than neg_ratio unseen items (the
 * reference would loop forever there). */
int crec_sample_pairwise(crec_ctx *ctx, const int64_t *ptr, const int32_t *items, int64_t n_users,
                         const int32_t *user_ids, int32_t item_nums, int neg_ratio, int32_t *u, int32_t *i, int32_t *j,
                         int64_t *n_out);
int crec_sample_pointwise(crec_ctx *ctx, const int64_t *ptr, const int32_t *items, int64_t n_users,
                          const int32_t *user_ids, int32_t item_nums, int neg_ratio, int32_t *u, int32_t *i, float *y,
                          int64_t *n_out);
int crec_sample_cml(crec_ctx *ctx, const int64_t *ptr, const int32_t *items, int64_t n_users, const int32_t *user_ids,
                    int32_t item_nums, int neg_ratio, int32_t *u, int32_t *i, int32_t *neg, int64_t *n_out);

/* SBPR's sampler (utils/sampler.py:102-141): per training positive of a user that has social items
 * (SPu = items the user's friends consumed and the user did not, utils/tools.py:116-127) and repetition: a social item
 * drawn from the user's SPu list, then a negative outside train and SPu items; spu_ptr / spu_items: the SPu lists as a CSR
 * parallel to the ui_train CSR (users without SPu: empty rows, skipped like `if u not in SPu: continue`), IN THE
 * REFERENCE'S LIST ORDER (the index of the drawn element is what the stream decides); spu_suk: per SPu entry the number
 * of the user's friends who consumed it (the social coefficient suk, :122-129, a property of (user, item)).
 * Capacity n_pos*neg_ratio each; suk as float (the dtype of the `suk` placeholder, SBPR.py:26). */
int crec_sample_sbpr(crec_ctx *ctx, const int64_t *ptr, const int32_t *items, int64_t n_users, const int32_t *user_ids,
                     int32_t item_nums, int neg_ratio, const int64_t *spu_ptr, const int32_t *spu_items, const int32_t *spu_suk,
                     int32_t *u, int32_t *i, int32_t *i_s, int32_t *i_neg, float *suk, int64_t *n_out);

/* ---- the SAME samplers on the device (bit-exact: stream words, index streams and final generator state) ---------
 * replaces pairwise_ranking_sampler / pointwise_ranking_sampler / ranking_sampler_cml (utils/sampler.py:46-74, 10-43,
 * 77-99) as called at the top of every epoch (model/RankingRecommender.py:37,49,92), with the columns left in HBM
 * where the train step reads them.  The MT19937 word stream itself is produced on the GPU by jump-ahead chains,
 * randint's masked rejection is a stream compaction, the per-user redraw loop and the shuffle's draws are tile
 * walks by one workgroup each, the permutation is a sort + pointer chase (csrc/mt_sampler.hip).
 *   crec_dsampler_create: ui_train as in crec_sample_* [host arrays]; builds the device copy + per-user item bitmaps.
 *       CREC_EUNSUPPORTED when item_nums > 262144 or users x items / 8 > 2 GiB (the host samplers take those).
 *   crec_dsampler_sample: ENQUEUES one epoch's sample on `stream` from the generator state (key, pos) =
 *       np.random.get_state()[1:3].  kind 0 pairwise: u, i, third = j int32 [n_pos*neg_ratio];  1 pointwise: u, i,
 *       third = y float [n_pos*(1+neg_ratio)];  2 cml: u, i [n_pos], third = neg int32 [n_pos, neg_ratio]  [dev].
 *   crec_dsampler_finish: waits for it; *n_out instances; (key_out, *pos_out) = the generator state afterwards, to
 *       be put back with np.random.set_state.  (If the epoch rejected more draws than the stream budget sized from
 *       the expected rejection rate allowed, the sample is redone with twice the budget -- same result.)
 *   crec_dsampler_stats: stream words used by the negatives / by the shuffle, candidates produced / used, then tile
 *       iterations, shader clocks and 100 MHz ticks of the negatives walk and of the shuffle-target walk, then how often the
 *       parallel negatives phase handed a sample to the walk, and why / where the last time. */
typedef struct crec_dsampler crec_dsampler;
int crec_dsampler_create(crec_ctx *ctx, const int64_t *ptr, const int32_t *items, int64_t n_users, const int32_t *user_ids,
                         int32_t item_nums, crec_dsampler **out, void *stream);
void crec_dsampler_destroy(crec_dsampler *s);
int crec_dsampler_sample(crec_ctx *ctx, crec_dsampler *s, int kind, int neg_ratio, const uint32_t key[624], int32_t pos,
                         int32_t *u, int32_t *i, void *third, void *stream);
int crec_dsampler_finish(crec_ctx *ctx, crec_dsampler *s, int64_t *n_out, uint32_t key_out[624], int32_t *pos_out);
int crec_dsampler_stats(const crec_dsampler *s, int64_t out[13]);
/* test hook [host]: the block of 624 raw MT19937 words that lies c*128 refills behind `key` (1 <= c <= 128), by the
 * jump polynomial the device chains use (csrc/mt_jump.cpp) */
int crec_debug_mt_jump(const uint32_t key[624], int c, uint32_t out_block[624]);

/* ---- device-side negative sampling (optional; NOT stream-compatible with the reference) --------------
 * Same acceptance rule as utils/sampler.py:18-31,50-66,82-93 -- neg_ratio items per training positive, uniform over
 * the catalogue, redrawn while seen by the user or already in the positive's group -- but drawn from a counter-based
 * generator keyed by (seed, epoch, positive, draw), one thread per positive, so an epoch costs one launch instead
 * of a serial MT19937 walk.  edge_user [dev][n_pos]: the user of every training positive in ui_train order;
 * seen_ptr/seen_items [dev]: CSR by user id, items sorted; max_seen [host]: the largest row length of that CSR
 * (termination check); neg [dev][n_pos][neg_ratio] out.  The epoch shuffle is the caller's (device randperm). */
int crec_sample_negatives_device(crec_ctx *ctx, uint64_t seed, uint64_t epoch, int64_t n_pos, const int32_t *edge_user,
                                 const int64_t *seen_ptr, const int32_t *seen_items, const int32_t *max_seen,
                                 int32_t item_nums, int32_t neg_ratio, int32_t *neg, void *stream);

/* ---- BPR training (device) ------------------------------------------------------------
 * replaces, per batch, sess.run([self.train, self.loss], {u_idx, i_idx, j_idx})
 * (model/RankingRecommender.py:46) on the graph of model/ranking/BPR.py:31-44 with
 * tf.train.AdamOptimizer (utils/tools.py:83-84). */
typedef struct {
    float *P, *Q;     /* [dev] [U,d], [I,d] parameters ('BPR_params/P', 'BPR_params/Q', BPR.py:28-29) */
    float *mP, *vP;   /* [dev] Adam slots of P */
    float *mQ, *vQ;   /* [dev] Adam slots of Q */
    int64_t U, I;
    int32_t d;
} crec_bpr_tables;

typedef struct {
    float lr, beta1, beta2, eps; /* AdamOptimizer(lr) defaults: 0.9, 0.999, 1e-8 */
    float reg;                   /* configs['reg'] (BPR.py:13) */
    int32_t adam_mode;           /* CREC_ADAM_* */
    /* optimizer state, updated by the call: fp32 running products like TF's beta1_power /
     * beta2_power slot variables.  Initialise to beta1, beta2 and step = 0. */
    float beta1_power, beta2_power;
    int64_t step;
} crec_adam;

/* One epoch's batches: contiguous slices [b*B, (b+1)*B) of the sampled triples (last one
 * partial), exactly the slicing of RankingRecommender.train_model (:39-47).
 *   u, i, j        [dev] int32 [N]
 *   loss_per_batch [dev] float [ceil(N/B)]: loss_b = sum softplus(-x) + reg*(l2 terms)
 * train_model's return value is mean(loss_per_batch) (:61).
 * B: the reference takes any batch_size (model/Recommender.py:17): 1 .. 524288 here.  Beyond 8192 the batch's index (the unique +
 * segment-sum bookkeeping) is built in global memory, the same index bit for bit (csrc/index_big.hip); adam_mode tf1 then runs the call
 * as deferred Adam with the flush at its end -- TF1's tables after every call, one whole-table sweep per call instead of one per step
 * (beta1 <= 0.9; refused above that, as crec_bpr_train_epoch_deferred is).  Tables whose rows and slots make 64 MB and more take the
 * same form at any batch size (a sweep per step would cost more than the step: 138 k x 27 k rows, d = 64: 56 -> 31 us per step;
 * beta1 > 0.9 keeps the sweep); option "tf1_sweep" = 1 sweeps regardless. */
int crec_bpr_train_epoch(crec_ctx *ctx, const crec_bpr_tables *t, crec_adam *opt, const int32_t *u, const int32_t *i,
                         const int32_t *j, int64_t N, int32_t B, float *loss_per_batch, void *stream);

/* adam_mode CREC_ADAM_DEFERRED: lastP [U], lastQ [I] int32 [dev], zero-initialised by the caller, hold the step at which
 * each row's (var, m, v) are current.  After an epoch the tables are NOT what TF1's would be until crec_bpr_adam_flush
 * has replayed the outstanding steps of every row: call it before evaluating, checkpointing or reading the tables
 * (replaces nothing in the reference: there every sess.run leaves all rows current). */
typedef struct {
    int32_t *lastP, *lastQ;
} crec_bpr_deferred;
int crec_bpr_train_epoch_deferred(crec_ctx *ctx, const crec_bpr_tables *t, const crec_bpr_deferred *df, crec_adam *opt,
                                  const int32_t *u, const int32_t *i, const int32_t *j, int64_t N, int32_t B,
                                  float *loss_per_batch, void *stream);
int crec_bpr_adam_flush(crec_ctx *ctx, const crec_bpr_tables *t, const crec_bpr_deferred *df, const crec_adam *opt, void *stream);

/* ---- pointwise NCF family: GMF / MLP / NeuMF (device) --------------------------------------------
 * replaces, per batch, sess.run([self.train, self.loss], {u_idx, i_idx, y}) (model/RankingRecommender.py:52-58)
 * on the graphs of model/ranking/GMF.py:29-57, MLP.py:31-77, NeuMF.py:33-105 (sigmoid cross-entropy,
 * utils/tools.py:68-69; l2 on the gathered rows; AdamOptimizer: embedding tables take the IndexedSlices
 * path and are swept on every row, the small parameters take dense ApplyAdam).
 * A crec_param is a dense row-major fp32 variable with its Adam slots, all caller-owned [dev]. */
typedef struct {
    float *var, *m, *v;
    int64_t rows;
    int32_t cols;
} crec_param;

/* configs['loss_func'] of the pointwise models: 'cross_entropy' = sum sigmoid_cross_entropy_with_logits(y, logits)
 * (utils/tools.py:68-69), 'square' = sum (y - logits)^2 (:74-75).  The other two names of get_loss cannot train these
 * graphs in the reference either: 'bpr' ignores the logits (no gradient), 'hinge' adds margin = None (TypeError). */
#define CREC_LOSS_CROSS_ENTROPY 0
#define CREC_LOSS_SQUARE 1

#define CREC_NCF_GMF 0
#define CREC_NCF_MLP 1
#define CREC_NCF_NEUMF 2

typedef struct {
    int32_t kind;            /* CREC_NCF_* */
    crec_param P_gmf, Q_gmf; /* [U,d], [I,d]: 'GMF_params/P','Q' or 'NeuMF_params/P_gmf','Q_gmf' (unused for MLP) */
    crec_param P_mlp, Q_mlp; /* [U,L0/2], [I,L0/2]: 'MLP_params/P','Q' or 'NeuMF_params/P_mlp','Q_mlp' (unused for GMF) */
    int32_t n_layers;        /* tower depth (<= 4); layer k maps width w_k = L0 / 2^k to w_k / 2 (MLP.py:34-37) */
    crec_param small;        /* ONE flat block [1, n]: W_0 | b_0 | W_1 | b_1 | ... | h, W_k row-major [w_k, w_k/2];
                                h = h_gmf [d] (GMF), h_mlp [last] (MLP), h_neumf [d + last] (NeuMF) */
    float reg_gmf, reg_mlp;  /* l2 weights of the gmf / mlp rows (configs 'reg' | 'reg_gmf' | 'reg1', 'reg_mlp' | 'reg2') */
    int32_t loss_func;       /* CREC_LOSS_*: get_loss(self.loss_func, self.y, self.logits), utils/tools.py:66-76 */
} crec_ncf;

/* u, i int32 [N], y float [N] [dev]: the shuffled instances of pointwise_ranking_sampler; batches are the
 * contiguous slices of RankingRecommender.train_model (:51-58); loss_per_batch [ceil(N/B)] [dev]. */
int crec_ncf_train_epoch(crec_ctx *ctx, const crec_ncf *model, crec_adam *opt, const int32_t *u, const int32_t *i,
                         const float *y, int64_t N, int32_t B, float *loss_per_batch, void *stream);

/* Full-catalogue logits of a GMF / MLP / NeuMF model: replaces sess.run(self.pre_scores) of the rs branch
 * (GMF.py:40,57, MLP.py:45-52,71-77, NeuMF.py:62-78,96-105).  logits [dev] float [Bt, I]; pre_scores is
 * sigmoid(logits), which ranks identically.  layers[0] <= 256. */
int crec_ncf_scores(crec_ctx *ctx, const crec_ncf *model, const int32_t *users, int32_t Bt, float *logits, void *stream);
/* np.argsort + seen filter (model/RankingRecommender.py:221-240) over a materialised score matrix
 * scores [dev] float [Bt, I], I <= 40000 per call (wider matrices: a stretch of the item axis at a time, runtime.topk_scores); outputs as crec_eval_topk. */
int crec_topk_scores(crec_ctx *ctx, const float *scores, int64_t I, const int32_t *users, int32_t Bt,
                     const int64_t *seen_ptr, const int32_t *seen_items, int32_t K, int32_t ascending, int32_t *topk_ids,
                     float *topk_scores, void *stream);

/* Leave-one-out / sampled-negative evaluation (RankingRecommender.test_model_loo, model/RankingRecommender.py:250-299):
 * only the user's candidate items compete -- the sampled negatives plus the held-out items that RankingPreprocess
 * stores in ui_test[u] (model/RankingPreprocess.py:121-129).  Same as crec_eval_topk / crec_topk_scores with the
 * list's meaning inverted: cand CSR by user id, items sorted ascending; an item NOT listed is never returned. */
int crec_eval_topk_candidates(crec_ctx *ctx, const float *P, const float *Q, int64_t I, int32_t d, const int32_t *users,
                              int32_t Bt, const int64_t *cand_ptr, const int32_t *cand_items, int32_t K,
                              int32_t ascending, int32_t *topk_ids, float *topk_scores, void *stream);
int crec_topk_scores_candidates(crec_ctx *ctx, const float *scores, int64_t I, const int32_t *users, int32_t Bt,
                                const int64_t *cand_ptr, const int32_t *cand_items, int32_t K, int32_t ascending,
                                int32_t *topk_ids, float *topk_scores, void *stream);

/* ---- Factorization Machine, rating path (device) ---------------------------------------------------
 * replaces sess.run([self.train, self.loss, self.y_pre]) / sess.run(self.y_pre) (model/RatingRecommender.py:50,73)
 * on the graph of model/rating/FM.py:20-53, and cal_rmse_mae (utils/metrics.py:22-29).
 *   w0 [1,1] ('fm_params/w0'), wi [N+1,1] ('wi'), vif [N+1,d] ('vif'); reg = configs['reg'].
 *   x_idx int32 [N,F], x_val float [N,F] (is_real_valued), y float [N]; F <= 64 (beyond 8 the row is walked twice instead of held); B*F <= 16384.
 *   The loss puts l2 on the FULL wi and vif tables (FM.py:52): both take dense ApplyAdam on every row. */
typedef struct {
    crec_param w0, wi, vif;
    float reg;
} crec_fm;
/* y_pre [N]: the pre-update predictions train_fm collects for the training RMSE (RatingRecommender.py:50-54) */
int crec_fm_train_epoch(crec_ctx *ctx, const crec_fm *model, crec_adam *opt, const int32_t *x_idx, const float *x_val,
                        const float *y, int64_t N, int32_t F, int32_t B, float *y_pre, float *loss_per_batch,
                        void *stream);
int crec_fm_predict(crec_ctx *ctx, const crec_fm *model, const int32_t *x_idx, const float *x_val, int64_t N, int32_t F,
                    float *y_pre, void *stream);
/* out2 [dev] double[2] = {rmse, mae} in float64 */
int crec_rmse_mae(crec_ctx *ctx, const float *y, const float *y_pre, int64_t n, double *out2, void *stream);

/* ---- Collaborative Metric Learning (device) ----------------------------------------------------------
 * replaces sess.run([self.train, self.loss], {u_idx, i_idx, neg_items}) (model/RankingRecommender.py:90-100)
 * on the graph of model/ranking/CML.py:34-70 (hinge on the closest sampled negative, WARP rank weight,
 * covariance regulariser over [Q;P]).  P ('cml_params/P') and Q take dense ApplyAdam on every row.
 *   u, i int32 [N]; neg int32 [N, K] (ranking_sampler_cml); B <= 65536. */
typedef struct {
    crec_param P, Q;
    float reg, margin; /* configs['reg'], configs['margin'] */
} crec_cml;
int crec_cml_train_epoch(crec_ctx *ctx, const crec_cml *model, crec_adam *opt, const int32_t *u, const int32_t *i,
                         const int32_t *neg, int64_t N, int32_t K, int32_t B, float *loss_per_batch, void *stream);
/* Evaluation form of CML._predict: the squared distance ranks like the ascending dot product of
 * P_aug [U, d+4] = [-2 p', 1, 0,0,0] and Q_aug [I, d+4] = [q, |q|^2, 0,0,0] (|p'|^2 is constant per user); feed them to
 * crec_eval_topk[_candidates] with ascending = 1 (cml_like, model/RankingRecommender.py:222-223).
 *   clip_users = 1: p' = clip_1(p_u), the rs branch (pre_scores from the re-bound u_embed, CML.py:76,84);
 *   clip_users = 0: p' = p_u, the loo / sampled-negatives branch (pre_scores = u_i_dist, built before _unit_clipping,
 *                   CML.py:42,81-82).  d <= 252. */
int crec_cml_eval_tables(crec_ctx *ctx, const float *P, const float *Q, int64_t U, int64_t I, int32_t d, int32_t clip_users,
                         float *P_aug, float *Q_aug, void *stream);

/* ---- Social BPR (device) -------------------------------------------------------------------
 * replaces sess.run([self.train, self.loss], {u_idx, i_idx, i_s_idx, i_neg_idx, suk}) (train_model_sbpr,
 * model/RankingRecommender.py:101-116) on the graph of model/ranking/SBPR.py:31-55: scores p.q + bias, two log-sigmoid
 * terms ((x_ui - x_uk)/suk and x_uk - x_uj), l2 on the gathered rows and biases, AdamOptimizer on IndexedSlices (P
 * 'sbpr_params/P', Q, and bias [item_nums + 1]: every row decays and moves each step).  u, i, k (social item), j int32 [N],
 * suk float [N] from crec_sample_sbpr; B <= 65536.  rs-evaluation is BPR's (pre_scores = u_embed @ Q^T, SBPR.py:57-61). */
typedef struct {
    crec_param P, Q, bias; /* bias: rows >= Q.rows, cols = 1 */
    float reg;
} crec_sbpr;
int crec_sbpr_train_epoch(crec_ctx *ctx, const crec_sbpr *model, crec_adam *opt, const int32_t *u, const int32_t *i,
                          const int32_t *k, const int32_t *j, const float *suk, int64_t N, int32_t B, float *loss_per_batch,
                          void *stream);

/* CUNE_BPR (model/ranking/CUNE_BPR.py:44-61): replaces sess.run([self.train, self.loss], {u_idx, i_idx, i_s_idx,
 * i_neg_idx}) of train_model_sbpr(is_suk=False) (CUNE_BPR.py:22, model/RankingRecommender.py:101-116).  SBPR's tables and
 * scores; the two log-sigmoid arguments are x_ui - x_uk and (x_uk - x_uj) / (s + 1.0), with s ('s', tf.Variable(0.0), :42) a
 * trainable scalar on the dense optimizer path (ApplyAdam).  u, i, k, j from crec_sample_sbpr (its suk column unused: the
 * stream does not depend on is_suk, utils/sampler.py:102-141).  The friends / SPu lists the sampler needs come from
 * get_topK_friends_and_SPu (utils/tools.py:130-207), host preprocessing outside this library. */
typedef struct {
    crec_param P, Q, bias; /* as crec_sbpr */
    crec_param s;          /* rows = cols = 1 */
    float reg;
} crec_cune;
int crec_cune_train_epoch(crec_ctx *ctx, const crec_cune *model, crec_adam *opt, const int32_t *u, const int32_t *i,
                          const int32_t *k, const int32_t *j, int64_t N, int32_t B, float *loss_per_batch, void *stream);

/* ---- Latent Relational Metric Learning (device) -----------------------------------------------
 * replaces sess.run([self.train, self.loss], {u_idx, i_idx, j_idx}) (model/RankingRecommender.py:39-47) on the graph of
 * model/ranking/LRML.py:36-63 -- joint = p_u * q, a = softmax(joint K), r = a M, dist = |p_u + r - q|^2,
 * loss = sum max(dist_i - dist_j + margin, 0) + reg * l2 of the gathered rows -- with AdamOptimizer (P, Q: IndexedSlices,
 * every row decays and moves; K [d, mem], M [mem, d]: dense), and sess.run(self.pre_scores) (LRML.py:70-76).
 *   small: ONE flat block [K | M] (K row-major [d][mem], then M row-major [mem][d]) with its Adam slots, rows * cols =
 *          2 * d * mem_size.  embed_size a multiple of 4, at most 128; mem_size <= 64; B <= 65536.
 *   crec_lrml_train_epoch: u, i, j int32 [N] from crec_sample_pairwise; loss_per_batch [ceil(N/B)] summed losses.
 *   crec_lrml_scores: scores [Bt][I] = the squared distance of every (user, item) pair; rank ascending (cml_like,
 *          model/RankingRecommender.py:222-223) with crec_topk_scores[_candidates]. */
typedef struct {
    crec_param P, Q, small;
    int32_t mem_size;
    float reg, margin;
} crec_lrml;
int crec_lrml_train_epoch(crec_ctx *ctx, const crec_lrml *model, crec_adam *opt, const int32_t *u, const int32_t *i,
                          const int32_t *j, int64_t N, int32_t B, float *loss_per_batch, void *stream);
int crec_lrml_scores(crec_ctx *ctx, const crec_lrml *model, const int32_t *users, int32_t Bt, float *scores, void *stream);

/* ---- Translational Collaborative Filtering (device) -------------------------------------------
 * replaces sess.run([self.train, self.loss], {u_idx, i_idx, j_idx}) (model/RankingRecommender.py:39-47) on the graph of
 * model/ranking/TransCF.py:39-77 -- NU = A_ui Q, NI = A_iu P (row-normalised training matrix and its transpose,
 * utils/tools.py:100-113), dist = |p_u + NU[u] * NI[i] - q_i|^2, hinge on dist_i - dist_j plus the neighbourhood (reg1) and
 * distance (reg2) regularisers -- with AdamOptimizer; both tables receive dense gradients (the sparse-dense products reach
 * every row), i.e. the dense ApplyAdam kernel on every row.
 *   ui_ptr [U+1] / ui_items: a user's training items, by user id; iu_ptr [I+1] / iu_users: an item's training users, by
 *   item id (device pointers; the same interaction set both ways).  u, i, j int32 [N] from crec_sample_pairwise; B <= 65536.
 *   crec_transcf_scores: scores [Bt][I] = |a + NU[u] * NI[item] - q_item|^2 with a = clip_1(p_u) (clip_users = 1: the rs
 *   branch of _predict, TransCF.py:88-91) or p_u (0: the loo / sampled-negatives branch, ui_dist, :86-87); nbr_ws: device
 *   scratch of (U + I) * d floats (NU, NI of the current tables).  Rank ascending (cml_like) with crec_topk_scores. */
typedef struct {
    crec_param P, Q;
    const int64_t *ui_ptr;
    const int32_t *ui_items;
    const int64_t *iu_ptr;
    const int32_t *iu_users;
    float reg1, reg2, margin;
} crec_transcf;
int crec_transcf_train_epoch(crec_ctx *ctx, const crec_transcf *model, crec_adam *opt, const int32_t *u, const int32_t *i,
                             const int32_t *j, int64_t N, int32_t B, float *loss_per_batch, void *stream);
int crec_transcf_scores(crec_ctx *ctx, const crec_transcf *model, const int32_t *users, int32_t Bt, int32_t clip_users,
                        float *nbr_ws, float *scores, void *stream);

/* ---- data-parallel training over RCCL (one process per GPU of one node) -----------------------
 * The path shards with one real exchange: gradients of the same Q row must be summed over all ranks'
 * triples before Adam.  Users (rows of P, their slots, their triples) are sharded over ranks, Q and its
 * slots are replicated; every step each rank emits its dense partial dQ, one ncclAllReduce sums it,
 * and every replica applies the identical tf1 Adam update (replicas stay bit-identical).
 *   crec_comm_unique_id : rank 0 creates the 128-byte RCCL id; the caller ships it to the other ranks
 *   crec_comm_init      : collective over all ranks (ncclCommInitRank on the ctx's device)
 *   crec_bpr_train_epoch_dp : like crec_bpr_train_epoch on this rank's shard (t->P = this rank's user
 *       rows, u = row ids inside the shard, t->Q = the replica).  Collective: every rank must pass the
 *       same number of batches ceil(N/B).  loss_per_batch receives the sum over ranks.  adam_mode tf1. */
#define CREC_UNIQUE_ID_BYTES 128
int crec_comm_unique_id(crec_ctx *ctx, void *id_out);
int crec_comm_init(crec_ctx *ctx, int rank, int world, const void *id);
int crec_comm_destroy(crec_ctx *ctx);
int crec_bpr_train_epoch_dp(crec_ctx *ctx, const crec_bpr_tables *t, crec_adam *opt, const int32_t *u, const int32_t *i,
                            const int32_t *j, int64_t N, int32_t B, float *loss_per_batch, void *stream);

/* ---- row-sharded training over the GPUs of one node (device) ------------------------------------------
 * No reference counterpart: the reference is one process on one device (main.py:40-47).  This is the form of the
 * BPR step for tables that outgrow a GPU (BASELINE.json configs[4]: 10 M users x 1 M items, d = 128):
 *   users -- rows of P, their slots, their triples -- are range-sharded: all P traffic is local;
 *   items are dealt round-robin: item g is local row g / G of rank g % G, whose Q shard holds ceil(I / G) rows;
 *   per batch: the referenced rows travel to the ranks that reference them (all-to-all-v), every rank runs the lazy
 *   BPR step on its own triples against those rows, the per-row gradient sums travel back to their owners
 *   (all-to-all-v), and the owner adds them in rank order and applies lazy Adam.  The row-id lists are exchanged
 *   once per chunk of batches, not per step.  No dense all-reduce of a table.
 * A global batch of G * B triples processed this way equals the single-GPU lazy step on the concatenated batch up
 * to the order of the fp32 additions (a row's sum is formed per rank first, then over ranks in rank order).
 *
 * Communicator: crec_comm_init (RCCL, one process per GPU) or crec_comm_init_local (the ranks are `world` ctxs of
 * THIS process, each driven by its own host thread -- one per GPU of a node, or several ranks on one GPU in tests);
 * a ctx with neither is a world of one.
 *   t->P, slots   [U_local, d]   this rank's user rows; u = row ids inside the shard
 *   t->Q, slots   [>= ceil(I_global / G), d]   this rank's item shard; i, j = GLOBAL item ids
 *   collective: every rank passes the same number of batches ceil(N / B), N >= 1; adam_mode lazy; B <= 524288 (beyond 8192 the
 *   plan, the owner-side index and the compact-id index are built in global memory; the transport's window must hold a batch's
 *   compact table, 2 * B * d * 4 bytes per half).
 *   loss_per_batch [dev] receives the sum over ranks.  stats (optional) accumulates what this rank sent to OTHER ranks. */
typedef struct {
    int64_t bytes_ids_sent, bytes_rows_sent, bytes_grads_sent; /* to other ranks */
    int64_t rows_requested, rows_served;                       /* unique (batch, item) pairs asked for / served, incl. own */
    int64_t max_rows_served_per_batch;
    int64_t steps;
} crec_shard_stats;
int crec_comm_init_local(crec_ctx **ctxs, int world);
int crec_bpr_train_epoch_sharded(crec_ctx *ctx, const crec_bpr_tables *t, int64_t I_global, crec_adam *opt, const int32_t *u,
                                 const int32_t *i, const int32_t *j, int64_t N, int32_t B, float *loss_per_batch,
                                 crec_shard_stats *stats, void *stream);
/* Peer transport for both multi-GPU forms (csrc/crec_peer.hip): instead of grouped ncclSend / ncclRecv, every rank maps a window of
 * every other rank (HIP IPC over xGMI; needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this platform) and ONE kernel per exchange copies the
 * rank's segments into its own window, publishes them with a flag per peer and reads the peers' segments with plain loads.
 *   crec_comm_peer_create   allocates this rank's window (two halves used alternately: an exchange may carry window_bytes / 2 per
 *                           rank) and signal words, returns the CREC_PEER_BLOB_BYTES the OTHER ranks need (ship them with whatever
 *                           the application has: torch.distributed all_gather in cleverrec_amd/parallel.py)
 *   crec_comm_peer_connect  blobs = world x CREC_PEER_BLOB_BYTES in rank order; maps the peers.  World <= 16.
 *   crec_comm_peer_connect_local  the ranks are `world` ctxs of THIS process (tests: several ranks on one GPU; or one thread per GPU)
 *   crec_comm_set_timeout   how long a rank waits for a peer's flag before the communicator reports an error (default 20 s)
 *   crec_comm_status        synchronises the stream and reports such an error.  crec_bpr_train_epoch_sharded checks once per chunk
 *                           of batches (it reads counts on the host there anyway); crec_bpr_train_epoch_dp never drains the
 *                           stream: it sends the error word to the host behind each chunk and reports what earlier chunks or
 *                           calls brought back -- call crec_comm_status after the last epoch (cleverrec_amd/parallel.py does, in
 *                           DpBprState.synchronize)
 * crec_bpr_train_epoch_dp and crec_bpr_train_epoch_sharded then run over it unchanged.  Exchanges are collective and must be
 * enqueued on ONE stream per ctx.  crec_debug_alltoallv / crec_debug_allreduce: one exchange of the ctx's communicator (any
 * transport) on device buffers -- counts and displacements in 4-byte elements, host arrays of `world` entries (tests, tools/micro). */
#define CREC_PEER_BLOB_BYTES 192
int crec_comm_peer_create(crec_ctx *ctx, int64_t window_bytes, void *blob_out);
int crec_comm_peer_connect(crec_ctx *ctx, int rank, int world, const void *blobs);
int crec_comm_peer_connect_local(crec_ctx **ctxs, int world, int64_t window_bytes);
int crec_comm_set_timeout(crec_ctx *ctx, double seconds);
/* on & 1: window traffic as plain accesses between system-scope release / acquire fences instead of written-through system-scope
 * accesses (same results; the two forms are correct against each other).  on & 2: every all-reduce in ONE shot (every rank reads all operands)
 * instead of two from three ranks and 64 KiB on (a rank adds one slice, the slices are gathered) -- same sums in the same order; THIS
 * choice must be the same on every rank. */
int crec_comm_peer_fences(crec_ctx *ctx, int on);
int crec_comm_status(crec_ctx *ctx, void *stream);
int crec_debug_alltoallv(crec_ctx *ctx, const void *send, const int64_t *scnt, const int64_t *sdsp, void *recv, const int64_t *rcnt,
                         const int64_t *rdsp, void *stream);
int crec_debug_allreduce(crec_ctx *ctx, float *buf, int64_t n, void *stream);
/* The index work of that step for ONE batch, twice: the host form (csrc/shard_host.cpp: the planner the multi-process CPU
 * tests run over gloo, all pointers host) and, as test hooks, the device kernels the epoch launches (all pointers device except
 * srv_cnt) -- the GPU tests hold the second to the first.
 *   i, j [nb] global item ids -> req_keys [<= 2 nb] the batch's unique owner-major keys (g % G) * S + g / G ascending,
 *   n_req their number, req_off [G+1] where every owner's keys start, ci / cj [nb] every slot's rank in that list (its row
 *   of the compact table), ids_send [n_req] the request lists as local row ids, owner after owner.
 *   ids_recv: the peers' request lists, peer after peer (srv_cnt [G] rows each) -> order [n] the slots sorted by row
 *   (stable: peers in rank order), urows / useg [n_urows (+1)] the unique rows and where their slots start in `order`. */
int crec_shard_plan_host(const int32_t *i, const int32_t *j, int32_t nb, int32_t G, int64_t I_global, int32_t *req_keys, int32_t *n_req,
                         int32_t *req_off, int32_t *ci, int32_t *cj, int32_t *ids_send);
int crec_shard_owner_index_host(const int32_t *ids_recv, const int32_t *srv_cnt, int32_t G, int32_t *order, int32_t *urows, int32_t *useg,
                                int32_t *n_urows);
int crec_debug_shard_plan(crec_ctx *ctx, const int32_t *i, const int32_t *j, int32_t nb, int32_t G, int64_t I_global, int32_t *req_keys,
                          int32_t *n_req, int32_t *req_off, int32_t *ci, int32_t *cj, int32_t *ids_send, void *stream);
int crec_debug_shard_owner_index(crec_ctx *ctx, const int32_t *ids_recv, const int32_t *srv_cnt, int32_t G, int64_t S, int32_t *order,
                                 int32_t *urows, int32_t *useg, int32_t *n_urows, void *stream);
/* test hook: the index the lazy-family BPR steps (adam_mode lazy / deferred, SGD, Adagrad) read -- TF1's unique + segment-sum
 * bookkeeping of a batch (the IndexedSlices aggregation behind model/ranking/BPR.py:31-44) -- of the ceil(N / B) batches of a stream,
 * built by the in-LDS sort (big = 0; B <= 8192) or by the global-memory sort that larger batches take (big = 1): the two must agree
 * bit for bit.  All arrays are device pointers; u_desc / i_desc / j_desc int32 [nb][B][4], slot_pu int32 [nb][B][2], nslots int32 [nb][2]. */
int crec_debug_lazy_index(crec_ctx *ctx, const int32_t *u, const int32_t *i, const int32_t *j, int64_t N, int32_t B, int64_t U, int64_t I,
                          int32_t d, int32_t big, int32_t *u_desc, int32_t *i_desc, int32_t *j_desc, int32_t *slot_pu, int32_t *nslots,
                          void *stream);

/* ---- full-catalogue evaluation (device) -------------------------------------------------
 * replaces sess.run(self.pre_scores) (BPR._predict rs branch, BPR.py:51) + np.argsort +
 * the seen-item filter loop (model/RankingRecommender.py:221-240).
 *   users        [dev] int32 [Bt] row ids into P
 *   seen_ptr     [dev] int64 [U+1], seen_items [dev] int32: CSR of ui_train by user id, each
 *                user's items sorted ascending (empty row = user absent from ui_train)
 *   topk_ids     [dev] int32 [Bt,K], topk_scores [dev] float [Bt,K]: best first; ties by
 *                lower item id; -1 / -inf padding when fewer than K unseen items exist
 *   ascending    0: larger score is better; 1: smaller is better (cml_like, :222-223)
 *   K            1 .. 128 (the reference takes any topk, Recommender.py:27); beyond 32 the list is ranked 32 at a time,
 *                every pass with the ids ranked so far out of the competition (same result as one pass) */
int crec_eval_topk(crec_ctx *ctx, const float *P, const float *Q, int64_t I, int32_t d, const int32_t *users,
                   int32_t Bt, const int64_t *seen_ptr, const int32_t *seen_items, int32_t K, int32_t ascending,
                   int32_t *topk_ids, float *topk_scores, void *stream);

/* replaces cal_ranking_metrics (utils/metrics.py:9-19) over a user batch, float64.
 *   ks [host] int32 [nk] (each <= K); test CSR by user id [dev]
 *   hr, mrr, ndcg [dev] double [nk, Bt] */
int crec_rank_metrics(crec_ctx *ctx, const int32_t *topk_ids, const int32_t *users, int32_t Bt, int32_t K,
                      const int32_t *ks, int32_t nk, const int64_t *test_ptr, const int32_t *test_items, double *hr,
                      double *mrr, double *ndcg, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CREC_H */
