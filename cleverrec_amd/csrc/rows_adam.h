// Internal interface of rows_adam.hip (not part of the C ABI).
#pragma once
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
    float lr_t, beta1, beta2, eps;
};

struct crec_rows_index_view {
    int32_t *order; // [n_batches][per_batch] slot ids sorted by (row, slot)
    int2 *rowseg;   // [n_batches][rows] {first sorted position, count}
    int32_t per_batch;
    int64_t rows;
};

// one table's share of a merged parameter-update launch (crec_apply_all)
struct crec_rows_job {
    crec_rows_table t;
    const int2 *rowseg;      // this batch's {first sorted position, count} per table row
    const int32_t *order;    // this batch's sorted slot ids
    const float *contrib;    // [slots][d] contribution rows
    const float *dense_grad; // optional (see rows_adam.hip)
    float *row_out;          // optional
    crec_rows_opt o;
};

struct crec_small_job { // dense ApplyAdam of a small vector from partial sums
    float *var, *m, *v;
    const float *partial;
    int n_partials;
    int64_t stride;
    int n;
    crec_rows_opt o;
};

constexpr int CREC_APPLY_MAX_TABLES = 4;
struct crec_apply_set {
    crec_rows_job tables[CREC_APPLY_MAX_TABLES];
    int n_tables;
    int block_end[CREC_APPLY_MAX_TABLES]; // filled by crec_apply_all
    crec_small_job small;                 // small.n == 0: none
};

size_t crec_rows_index_bytes(int64_t n_batches, int32_t per_batch, int64_t rows);
int crec_rows_index(crec_ctx *ctx, const int32_t *idx, int64_t n_total, int32_t per_batch, int64_t rows, void *ws,
                    crec_rows_index_view *out, hipStream_t st);
int crec_rows_apply(crec_ctx *ctx, const crec_rows_table &t, const crec_rows_index_view &ix, int64_t batch,
                    const float *contrib, const float *dense_grad, const crec_rows_opt &o, hipStream_t st,
                    float *row_out = nullptr);
int crec_small_adam_dense(crec_ctx *ctx, float *var, float *m, float *v, const float *partial, int n_partials,
                          int64_t stride, int n, const crec_rows_opt &o, hipStream_t st);
// all tables (same row-width class) and the small vector of one step in a single launch
int crec_apply_all(crec_ctx *ctx, crec_apply_set &a, hipStream_t st);
