// Internal declarations shared by the translation units of libcrec.so.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/crec.h"

#define CREC_VERSION 100

struct crec_mt19937 {
    uint32_t key[624];
    int32_t pos;
};

// A device buffer the ctx grows on demand and keeps across calls.
struct crec_devbuf {
    void *ptr = nullptr;
    size_t bytes = 0;
};

// Optional per-kernel timing with HIP events on the launch stream (crec_profile_*).
enum crec_kernel_class {
    CREC_K_INDEX = 0,
    CREC_K_STEP_FUSED,
    CREC_K_USER_SIDE,
    CREC_K_STEPS_FLOW, // bpr_steps_flow: up to 64 tf1 steps per launch (`units` counts the steps)
    CREC_K_REDUCE_LOSS,
    CREC_K_EVAL_SWEEP,
    CREC_K_EVAL_MERGE,
    CREC_K_METRICS,
    CREC_K_FINISH,
    CREC_K_ADAM_DENSE,
    CREC_K_ROWS_INDEX,
    CREC_K_ROWS_APPLY,
    CREC_K_SMALL_ADAM,
    CREC_K_NCF_TILE,
    CREC_K_FM_INSTANCES,
    CREC_K_CML_INSTANCES,
    CREC_K_NCF_SCORES,
    CREC_K_SHARD_PLAN,
    CREC_K_SHARD_SERVE,
    CREC_K_SHARD_APPLY,
    CREC_K_PEER_ADAM, // peer_allreduce_adam: the replicated-Q step's exchange + dense Adam
    CREC_K_COUNT
};

struct crec_profile {
    bool on = false;
    struct rec {
        int cls;
        hipEvent_t a, b;
        bool ext; // a / b were handed to hipExtLaunchKernelGGL: they carry the dispatch's own begin / end timestamps
        bool skip; // launched while its predecessor was still running: its events also span the wait, not counted
        int64_t units; // see crec_profile::units
    };
    int64_t next_units = 1; // units of the next crec_prof_begin
    bool skip_next = false; // the next crec_prof_begin opens such a record
    bool open = false; // between crec_prof_begin and crec_prof_end
    std::vector<rec> pending;
    std::vector<hipEvent_t> pool;
    double total_ms[CREC_K_COUNT] = {0};
    int64_t launches[CREC_K_COUNT] = {0};
    int64_t units[CREC_K_COUNT] = {0}; // work units of the counted launches where a launch holds a varying number (steps of bpr_steps_flow)
};

struct crec_ctx {
    int device = -1;
    crec_profile prof;
    char err[512] = {0};
    crec_mt19937 rng;
    // BPR training workspace (bpr_train.hip)
    crec_devbuf idx_ws;   // per-batch inverted indices of one chunk of batches
    crec_devbuf big_ws;   // index_big.hip: sort buffers, run tables and scans of one chunk of large batches
    crec_devbuf idx_ws2;  // ... whose chunks are indexed on a side stream while the previous chunk's steps run: the second index buffer,
    hipStream_t idx_stream = nullptr; // the side stream, and the events that order the two streams ([0..1] index of a buffer ready,
    hipEvent_t idx_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr}; // [2..3] steps on a buffer done, [4] the caller's stream on entry)
    crec_devbuf contrib;  // lazy mode: contribution rows, heavy-row partials and ticket counters (setup_lazy_bufs)
    crec_devbuf dq;       // data-parallel step: dense per-item gradient sums [I, d]
    void *comm = nullptr; // ncclComm_t (RCCL), set by crec_comm_init
    int rank = 0, world = 1;
    struct crec_xch *xch = nullptr; // exchange object of the row-sharded path (crec_comm.h): local group, RCCL or world of one
    crec_devbuf shard_flag;          // one float: the collective can-every-rank-go-on verdict of a chunk
    crec_devbuf shard_ws, shard_ws2; // row-sharded step: plan side (request lists, compact ids, Qc / dQc), serve side
    int64_t *pinned64 = nullptr;    // page-locked host staging of the sharded path's counts
    size_t pinned64_n = 0;
    crec_devbuf shadow;   // second buffer set of the tf1-mode ping-pong: P, mP, vP, Q, mQ, vQ
    crec_devbuf loss_ws;  // per-triple loss terms of one chunk
    crec_devbuf loss_parts; // bpr_reduce_loss: partial sums and tickets per batch
    crec_devbuf lazy_wpart; // lazy_window: polled partials of front rows, all bits set between launches
    bool lazy_wpart_dirty = false; // (a launch gave up: entries may hold partials nobody took)
    crec_devbuf tf1_last;   // tf1 beyond 8192 triples per batch (run as deferred Adam + flush): the rows' last-step stamps of the call
    crec_devbuf misc;     // heavy-row partials + ticket counters
    char *tickets_at = nullptr; // where the counters were last cleared (they are re-zeroed by the kernels after use)
    size_t tickets_bytes = 0;
    // deferred Adam: lr_s of every step so far (host copy, device copy, the running beta powers behind it)
    std::vector<float> lr_hist;
    crec_devbuf lr_hist_dev;
    size_t lr_hist_uploaded = 0;
    float lr_hist_key[3] = {0, 0, 0}, lr_hist_b1p = 0, lr_hist_b2p = 0;
    char *lazy_tickets_at = nullptr; // the lazy step's counters (inside `contrib`), likewise
    size_t lazy_tickets_bytes = 0;
    size_t index_lds_set = 0;   // dynamic LDS limit already granted to bpr_index_batches
    int32_t *pinned = nullptr; // page-locked host buffer for the per-chunk counts read back from the index kernel
    size_t pinned_ints = 0;
    int32_t *pinned_dev = nullptr; // the same buffer as the device addresses it (null: not mapped, the host copies instead)
    int32_t *idx_done = nullptr;   // device: ticket counter of the index launch's workgroups
    int32_t idx_seq = 0;           // sequence number of the last index launch that published its counts itself
    // eval workspace (eval_topk.hip)
    crec_devbuf eval_ws;
    crec_devbuf ks_dev;
    int num_cus = 0;
    // tuning / test options: read ONCE from the environment at crec_ctx_create (CREC_CH, CREC_WIDE, CREC_WIDE_ROUND,
    // CREC_EVAL_SHAPE, CREC_EVAL_NO_BITMAP), changed afterwards only through crec_ctx_set_option -- never getenv on a hot path
    int knob_ch = 0;             // occurrences per heavy-row chunk (0: the mode's default)
    int knob_wide = -1;          // wide-group threshold of the tf1 step (-1: default 16, 0: off)
    int knob_wide_round = 0;     // experiment: pad the wide groups' workgroups to a multiple of this (0: chunk maximum)
    int knob_eval_shape = 0;     // 4 / 8: force an eval_sweep workgroup shape where legal (0: choose)
    int knob_eval_no_bitmap = 0; // 1: seen filter by binary search even where the bitmaps would fit
    int knob_no_zero_copy = 0;   // 1: the chunk's counts come back by copy + stream synchronisation (tests)
    int knob_shard_unfused = 0;  // 1: the row-sharded step over the peer transport as five launches (serve, exchange, step, exchange, apply) like over RCCL (tests)
    int knob_no_call_tail = 0;   // 1: a call that ends in the second buffer set brings its tables home in a launch of its own (default: inside the last chunk's loss reduction launch)
    int knob_tf1_sweep = 0;      // 1: adam_mode tf1 sweeps every row at every step whatever the tables' size (default: tables of 64 MB and more run the call as deferred Adam + flush)
    int knob_idx_no_ids16 = 0;   // 1: the tf1 index gathers its records' ids from the streams (test: the same records as from LDS)
    int knob_idx_overlap = 0;    // 1: chunks of batches are indexed a chunk ahead on a side stream, beside the previous chunk's steps (index_big's, and the lazy family's LDS-built ones)
    int knob_idx_one_wg = 0;     // 1: the tf1 step's index with ONE workgroup per (batch, side) as before round 3 (tests)
    int knob_dp_unfused = 0;     // 1: the replicated-Q step over the peer transport as all-reduce + separate Adam launch (tests)
    int knob_big_index = 0;      // 1: the lazy step's index through index_big.hip (global-memory sort) even for batches the LDS sort holds (tests)
    int knob_ncf_generic = 0;    // 1: the generic ncf_tile even for the shapes that have a specialised one (tests)
    int knob_dsampler_budget = 100; // percent of the device sampler's stream budget to start with (tests: < 100 forces the retry path)
    int knob_flow = -1;          // tf1 step: -1 / 1 all steps of a chunk in one dataflow launch where the shape allows (bpr_steps_flow), 0 a launch per step
    int knob_lazy_window = -1;   // lazy-family steps: W > 1 consecutive batches per launch (bpr_triples_lazy<.., WIN>), 0 / 1 a launch per batch, -1 by table size
    crec_devbuf lazy_done;       // lazy_window: the finishers' stamps
    int32_t lazy_stamp = 0;      // ... and the ctx-wide launch count they are drawn from
    int knob_flow_abl = 0;       // timing ablations of bpr_steps_flow (FlowArgs::abl; wrong results)
    int knob_flow_steps = 0;     // steps per bpr_steps_flow launch (0: up to 64; 1: a launch per step through the same kernel -- no version history, tests)
    // bpr_steps_flow: the two version histories (alternating between launches), what they are laid out for, how many slots of each
    // hold published rows (the rest holds the sentinel), which one the next launch fills; the give-up flag (page-locked, mapped)
    crec_devbuf flow_hist;
    void *flow_hist_at = nullptr;
    int64_t flow_key[3] = {0, 0, 0}; // {steps per history + 1, floats per P version, floats per Q version}
    int flow_dirty[2] = {0, 0}, flow_cur = 0;
    int32_t *flow_err = nullptr, *flow_err_dev = nullptr;
    crec_devbuf flow_deal;              // the row dealing by weight: counts, row -> position and position -> row of both sides
    int64_t flow_deal_key[3] = {0, 0, 0}; // {U, I, rows per workgroup} it was made for (0: none yet)
    int knob_flow_deal = 1;             // 0: rows dealt round-robin by id (tests: the results do not depend on the dealing)
    size_t flow_deal_lds_set = 0;
    int flow_blocks_per_cu[8][2] = {{0}}; // occupancy of bpr_steps_flow<LPR, FULL> by (log2 LPR, FULL); 0: not asked yet
    int knob_dsampler_general = 0; // negatives phase of the device sampler: 0 parallel tables (walk where they do not apply), 2 the four-wave walk, 1 the general walk (tests)
};

int crec_fail(crec_ctx *ctx, int code, const char *fmt, ...);
// profiling brackets: no-ops unless crec_profile_enable(ctx, 1)
void crec_prof_begin(crec_ctx *ctx, int cls, hipStream_t st);
void crec_prof_end(crec_ctx *ctx, hipStream_t st);
// inside a begin/end bracket with profiling on: the bracket's two events, to be attached to the dispatch itself
bool crec_prof_events(crec_ctx *ctx, hipEvent_t *a, hipEvent_t *b);
// kernel launch inside a crec_prof_begin / crec_prof_end bracket.  Profiling on: the launch carries the bracket's
// events (hipExtLaunchKernelGGL), so the measured time is the dispatch's begin-to-end time -- the same timestamps
// rocprofv3 --kernel-trace reports -- not stream time between two separate event records.
#define CREC_KLAUNCH(ctx, kernel, grid, block, lds, st, ...)                                        \
    do {                                                                                            \
        hipEvent_t ea_, eb_;                                                                        \
        if (crec_prof_events(ctx, &ea_, &eb_))                                                      \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, st, ea_, eb_, 0, __VA_ARGS__);          \
        else                                                                                        \
            hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                          \
    } while (0)
int crec_prof_flush(crec_ctx *ctx, hipStream_t st);
int crec_reserve(crec_ctx *ctx, crec_devbuf *b, size_t bytes);
// the ctx's side stream and its five events (created on first use): work that does not depend on what the caller's stream is busy with
int crec_side_stream(crec_ctx *ctx, hipStream_t *side);

#define CREC_HIP(ctx, call)                                                                            \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return crec_fail((ctx), CREC_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                             __FILE__, __LINE__);                                                      \
    } while (0)

#define CREC_REQUIRE(ctx, cond, ...)                      \
    do {                                                  \
        if (!(cond))                                      \
            return crec_fail((ctx), CREC_EINVAL, __VA_ARGS__); \
    } while (0)

static inline int64_t crec_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Named ranges for rocprofv3 --marker-trace around the library's launches (the reference brackets its phases with the `timer`
// decorator, utils/tools.py:18-28): an epoch's chunks of batches, an evaluation, a sample.  Off unless CREC_ROCTX=1 is in the
// environment when the first range opens (the roctx library is then loaded with dlopen; without it the ranges are no-ops).
struct crec_range {
    explicit crec_range(const char *name);
    ~crec_range();
    bool on;
};
