// Internal interface of rows_adam.hip (not part of the C ABI).
#pragma once
#include <cmath>
#include "crec_internal.h"

enum { ROWS_SPARSE_TF1 = 0, ROWS_SPARSE_LAZY = 1, ROWS_DENSE = 2 };

struct crec_rows_table {
    float *var, *m, *v;
    int64_t rows;
    int32_t d;
};

struct crec_rows_opt {
    int kind;         // ROWS_*
    float reg_occ;    // l2 weight applied once per occurrence of a row (tf.nn.l2_loss on gathered rows)
    float dense_reg;  // ROWS_DENSE: l2 weight on the whole table (FM.py:52)
    float cov_coef;   // ROWS_DENSE, CML: 2*reg/n of the covariance term; dense_grad then holds the column means
    int row_norm;     // row_out[r] = |row r|^2 of the values before the update (FM's l2 loss term)
    float lr_t, beta1, beta2, eps;
    int opt;          // 0: Adam; CREC_OPT_SGD, CREC_OPT_ADAGRAD (utils/tools.py:79-87): lr_t then holds the plain learning rate,
                      // a sparse table's untouched rows stay as they are (scatter_sub / SparseApplyAdagrad on the unique rows),
                      // Adagrad's accumulator is the m slot (the caller initialises it to 0.1), the v slot is not used
};

// the step's optimizer out of a crec_adam (adam_mode tf1 = Adam | CREC_OPT_SGD | CREC_OPT_ADAGRAD) into a crec_rows_opt:
// opt, lr_t (Adam: lr*sqrt(1-b2^t)/(1-b1^t) with the fp32 running powers; else the plain lr), betas, eps.
// lazy_ok: the model's embedding tables are reached by gathers only and updated as plain row jobs, so adam_mode lazy may turn
// their kind (preset ROWS_SPARSE_TF1 by the caller) into ROWS_SPARSE_LAZY -- Adam's formula on the rows a batch touches, every
// other row and its slots left as they are (what BPR's lazy mode does; not TF1's results); dense variables keep ApplyAdam.
inline int crec_rows_optimizer(crec_ctx *ctx, const crec_adam *opt, crec_rows_opt *o, bool lazy_ok = false)
{
    const bool lazy = lazy_ok && opt->adam_mode == CREC_ADAM_LAZY && o->kind == ROWS_SPARSE_TF1;
    const bool adam = opt->adam_mode == CREC_ADAM_TF1 || lazy;
    CREC_REQUIRE(ctx, adam || opt->adam_mode == CREC_OPT_SGD || opt->adam_mode == CREC_OPT_ADAGRAD,
                 lazy_ok ? "this model's step implements adam_mode tf1 / lazy (Adam), CREC_OPT_SGD and CREC_OPT_ADAGRAD"
                         : "this model's step implements adam_mode tf1 (Adam), CREC_OPT_SGD and CREC_OPT_ADAGRAD");
    if (lazy)
        o->kind = ROWS_SPARSE_LAZY;
    o->opt = adam ? 0 : opt->adam_mode;
    o->lr_t = adam ? (opt->lr * sqrtf(1.0f - opt->beta2_power)) / (1.0f - opt->beta1_power) : opt->lr;
    o->beta1 = opt->beta1;
    o->beta2 = opt->beta2;
    o->eps = opt->eps;
    return CREC_OK;
}

// A batch's slots may be indexed in `chunks` equal runs of slots, each sorted by its own workgroup (a per-step index
// is then `chunks` short sorts side by side instead of one long one).  A row's slots are visited chunk by chunk, and
// inside a chunk in slot order -- i.e. still in ascending slot order overall.
struct crec_rows_index_view {
    int32_t *order; // [n_batches][per_batch] slot ids: per chunk, sorted by (row, slot)
    int2 *rowseg;   // [n_batches][chunks][rows] {first position in the batch's order array, count}
    int32_t per_batch;
    int64_t rows;
    int32_t chunks;
};

// one table's share of a merged parameter-update launch (crec_apply_all)
struct crec_rows_job {
    crec_rows_table t;
    int32_t chunks;          // index chunks of this batch (0 is read as 1)
    const int2 *rowseg;      // this batch's [chunks][rows] {first sorted position, count}
    const int32_t *order;    // this batch's sorted slot ids
    const float *contrib;    // [slots][d] contribution rows
    const float *dense_grad; // optional (see rows_adam.hip)
    float *row_out;          // optional
    crec_rows_opt o;
    // optional partner: a second table with the SAME rows, width and slots (rowseg / order), its own contribution rows and
    // l2 weight, updated by the same wave (rows_apply_row_pair); t2.var == nullptr: none
    crec_rows_table t2;
    const float *contrib2;
    float reg_occ2;
};

struct crec_small_job { // dense ApplyAdam of a small vector from partial sums
    float *var, *m, *v;
    const float *partial;
    int n_partials;
    int64_t stride;
    int n;
    crec_rows_opt o;
};

struct crec_scalar_job { // a [rows] vector treated as a one-column table (FM's wi): slot scalars summed per row,
    float *var, *m, *v;  // dense ApplyAdam with dense_reg * var; sq_out[r] = var[r]^2 before the update
    int64_t rows;        // rows == 0: none
    const int2 *rowseg;
    const int32_t *order;
    const float *contrib; // [slots]
    float *sq_out;
    crec_rows_opt o;
    int32_t chunks; // index chunks of this batch (0 is read as 1): rowseg is [chunks][rows]
};

constexpr int CREC_APPLY_MAX_TABLES = 4;
struct crec_apply_set {
    crec_rows_job tables[CREC_APPLY_MAX_TABLES];
    int n_tables;
    int block_end[CREC_APPLY_MAX_TABLES]; // filled by crec_apply_all
    crec_scalar_job scalar;               // after the tables: 256 rows per block
    int scalar_end;                       // filled by crec_apply_all
    crec_small_job small;                 // small.n == 0: none; its blocks lead the grid
    int small_blocks;                     // filled by crec_apply_all
};

size_t crec_rows_index_bytes(int64_t n_batches, int32_t per_batch, int64_t rows, int32_t chunks = 1);
int crec_rows_index(crec_ctx *ctx, const int32_t *idx, int64_t n_total, int32_t per_batch, int64_t rows, void *ws,
                    crec_rows_index_view *out, hipStream_t st, int32_t chunks = 1);
int crec_rows_apply(crec_ctx *ctx, const crec_rows_table &t, const crec_rows_index_view &ix, int64_t batch,
                    const float *contrib, const float *dense_grad, const crec_rows_opt &o, hipStream_t st,
                    float *row_out = nullptr);
int crec_small_adam_dense(crec_ctx *ctx, float *var, float *m, float *v, const float *partial, int n_partials,
                          int64_t stride, int n, const crec_rows_opt &o, hipStream_t st);
// all tables (same row-width class) and the small vector of one step in a single launch
int crec_apply_all(crec_ctx *ctx, crec_apply_set &a, hipStream_t st);
