// Shared by ncf.hip (training) and ncf_eval.hip (full-catalogue scoring): the device-side view of a crec_ncf.
#pragma once
#include "crec_internal.h"

constexpr int MAX_LAYERS = 4;

struct NcfDev {
    int kind; // 0 GMF, 1 MLP, 2 NeuMF
    const float *Pg, *Qg, *Pm, *Qm;
    const float *small; // [W_0 | b_0 | W_1 | b_1 | ... | h]
    int dg, n_layers, L0;
    int w_off[MAX_LAYERS], b_off[MAX_LAYERS], h_off, n_small;
    float reg_g, reg_m;
    long long *dbg; // -DNCF_STAMPS builds only: per-phase clock stamps of one tile (tools/ncf_variants.sh)
};

// validates a crec_ncf and lays out the flat small-parameter block [W_0 | b_0 | ... | h]
inline int ncf_describe(crec_ctx *ctx, const crec_ncf *mdl, NcfDev *d)
{
    CREC_REQUIRE(ctx, mdl->kind >= CREC_NCF_GMF && mdl->kind <= CREC_NCF_NEUMF, "ncf kind");
    const bool has_g = mdl->kind != CREC_NCF_MLP, has_m = mdl->kind != CREC_NCF_GMF;
    *d = NcfDev{};
    d->kind = mdl->kind;
    d->reg_g = mdl->reg_gmf;
    d->reg_m = mdl->reg_mlp;
    int off = 0;
    if (has_g) {
        CREC_REQUIRE(ctx, mdl->P_gmf.var && mdl->Q_gmf.var && mdl->P_gmf.cols == mdl->Q_gmf.cols, "gmf tables");
        d->Pg = mdl->P_gmf.var;
        d->Qg = mdl->Q_gmf.var;
        d->dg = mdl->P_gmf.cols;
        CREC_REQUIRE(ctx, d->dg % 4 == 0 && d->dg >= 4 && d->dg <= 256, "embed_size %d: need a multiple of 4 in [4, 256]", d->dg);
    }
    if (has_m) {
        CREC_REQUIRE(ctx, mdl->P_mlp.var && mdl->Q_mlp.var && mdl->P_mlp.cols == mdl->Q_mlp.cols, "mlp tables");
        CREC_REQUIRE(ctx, mdl->n_layers >= 1 && mdl->n_layers <= MAX_LAYERS, "1..4 MLP layers");
        d->Pm = mdl->P_mlp.var;
        d->Qm = mdl->Q_mlp.var;
        d->n_layers = mdl->n_layers;
        d->L0 = 2 * mdl->P_mlp.cols;
        CREC_REQUIRE(ctx, mdl->P_mlp.cols % 4 == 0 && d->L0 <= 512, "layers[0] %d: need a multiple of 8, at most 512", d->L0);
        int w = d->L0;
        for (int k = 0; k < d->n_layers; ++k) { // W_k [w, w/2], b_k [w/2]  (MLP.py:34-37)
            CREC_REQUIRE(ctx, w % 2 == 0 && w >= 2, "layer widths must halve");
            d->w_off[k] = off;
            off += w * (w / 2);
            d->b_off[k] = off;
            off += w / 2;
            w /= 2;
        }
        d->h_off = off;
        off += d->dg + w;
    } else {
        d->h_off = 0;
        off = d->dg;
    }
    d->n_small = off;
    CREC_REQUIRE(ctx, mdl->small.var && mdl->small.m && mdl->small.v && mdl->small.rows * mdl->small.cols == off,
                 "small parameter block must hold %d floats ([W_0|b_0|...|h])", off);
    d->small = mdl->small.var;
    return CREC_OK;
}
