// BPR training step for gfx950: gather -> pairwise loss -> per-row gradient sums -> Adam,
// without atomics and without ever materialising gradient rows in HBM.
//
// Replaces, per batch, sess.run([train, loss]) on the graph of model/ranking/BPR.py:31-44
// with tf.train.AdamOptimizer semantics (utils/tools.py:83-84); see include/crec.h.
//
// Per chunk of batches, one launch:
//   bpr_index_batches  one workgroup per (batch, side): stable radix sort in LDS of (row, position)
//                      pairs (block_sort.h) -> unique rows, segment starts, occurrence records {u, i, j, pos}.
//                      This is the "unique + unsorted_segment_sum" bookkeeping of TF1's sparse
//                      Adam, done once for the whole chunk, in batch position order, i-slices
//                      before j-slices, so every row sum has one fixed order (deterministic).
//                      tf1 mode also writes row -> {segment start, length} for every table row.
// Per batch:
//   tf1 mode   ONE launch, bpr_step_fused: every row of P and Q moves every step (TF1 decays
//              m, v and applies var -= lr_t*m/(sqrt(v)+eps) on all rows), so the step reads
//              the tables from one buffer set and writes the other (ping-pong): no gather can
//              see a row updated by the same step, and user-row and item-row waves run side by
//              side.  An item wave recomputes s_b = sigmoid(-x_b) of its occurrences from
//              p_u, q_i, q_j with the very instruction sequence the user wave uses, so both
//              sides use bit-identical s_b.
//   lazy mode  ONE launch in place, bpr_triples_lazy, a lane group per triple.  The index kernel tells every
//              triple, for each of its three rows, whether the row occurs once in the batch.  One round of
//              loads brings p_u, q_i, q_j and the Adam slots of the single-occurrence rows; those rows (nearly
//              all of them when the tables are large) are finished right there -- no contribution rows, no
//              second pass.  For a row that occurs several times every contributing group publishes its
//              contribution row at the slot's position in the side's sorted order and draws a ticket on the
//              row's counter; the group that draws the last one adds the row's contributions in sorted
//              (= batch) order and applies Adam.  Rows with more than CH occurrences do this per chunk of CH
//              slots first and then over the chunks' partial sums.  Nobody waits for anybody, every sum has
//              one fixed order, and a multi-occurrence row is only ever written by its last contributor,
//              after every reader of it has taken its ticket.
//   tf1 mode:   a row with more than CH occurrences in the batch is cut into chunks handled by separate
//              waves; the chunk wave that draws the row's last ticket sums the partials in chunk order and
//              applies Adam (chunk_wave), so there is no extra launch.
// Per chunk: bpr_reduce_loss sums the per-triple loss terms of each batch in a fixed order.
//
// A wave owns one table row.  It splits into G = 64/LPR groups of LPR lanes, each lane holding
// a float4 of a row (LPR*4 >= d): G occurrences are processed at once and every row access is
// one coalesced 16-byte-per-lane load.  The wave fetches up to 64 occurrence records with one
// coalesced load and hands them to its groups by cross-lane reads; row gathers of up to UNR
// sub-iterations are issued before the first is consumed.  Dots reduce with DPP adds.
//
// Compiled with -ffp-contract=off: each fp32 multiply/add is rounded separately, like the
// element-wise Eigen kernels TF1 runs (and like the CPU oracle), so the Adam arithmetic is
// bit-reproducible.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstring>
#include <queue>
#include "crec_internal.h"
#include "block_sort.h"
#include "lazy_desc.h"
#include "crec_comm.h"

#include <rccl/rccl.h>
typedef unsigned crec_u4v __attribute__((ext_vector_type(4)));

namespace {

constexpr int PREP_THREADS = 1024;
constexpr int MAX_B = 8192; // triples per batch bpr_index_batches sorts in LDS; larger batches: index_big.hip (lazy-family steps)
#ifndef CREC_UNR
#define CREC_UNR 2
#endif
#ifndef CREC_RPW
#define CREC_RPW 4
#endif
#ifndef CREC_PIPE
#define CREC_PIPE 1
#endif
constexpr int UNR = CREC_UNR;           // sub-iterations per trip
constexpr int ROWS_PER_WAVE = CREC_RPW; // table rows a normal wave owns (capped by the groups per wave)
constexpr bool PIPELINED = CREC_PIPE;   // two trips in flight
#ifndef CREC_ABL
#define CREC_ABL 0
#endif
#ifndef CREC_FLOW_STAMP_STEP
#define CREC_FLOW_STAMP_STEP 30
#endif
#ifdef CREC_STAMPS // debug build: every wave of bpr_step_fused records when it started and ended and when its loads
                   // arrived (tools/wave_stamps.sh); [wave][8] = {begin, end, shader clocks, xcd, marks 0..3}, 100 MHz clock
__device__ unsigned long long crec_stamps[8 * 16384];
extern "C" int crec_debug_stamps(unsigned long long *out, size_t n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(crec_stamps), n * sizeof(unsigned long long));
}
extern "C" int crec_debug_stamps_clear(void)
{
    static unsigned long long zeros[8 * 16384];
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(crec_stamps), zeros, sizeof(zeros));
}
// lazy_window's timeline (tools/window_stamps.py): per batch of the launch 16 words from crec_stamps[8 * 16000]; "min" events are
// kept as the maximum of (2^62 - t), the array starts at zero
constexpr int WIN_T_BASE = 8 * 16000;
__device__ __forceinline__ void win_t(int wb, int k, bool is_min)
{
    const unsigned long long t = wall_clock64();
    atomicMax(&crec_stamps[WIN_T_BASE + wb * 16 + k], is_min ? (1ull << 62) - t : t);
}
__device__ __forceinline__ void crec_mark(int slot, int dep) // the time at which `dep` is available to the wave
{
#ifndef CREC_STAMP_MARKS // marks make the wave wait for `dep` where it stands: they perturb what they measure
    return;
#endif
    asm volatile("" ::"v"(dep));
    const unsigned long long t = wall_clock64();
    const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0 && w < 16384)
        crec_stamps[8 * w + 4 + slot] = t;
}
#else
#define crec_mark(slot, dep)
#define win_t(wb, k, is_min)
#endif
#ifndef CREC_FUSED_BLOCK
#define CREC_FUSED_BLOCK 256 // threads per workgroup of bpr_step_fused
#endif
#ifndef CREC_ITEMS_FIRST
#define CREC_ITEMS_FIRST 0 // 1: item-row waves take the low wave ids (they are the longer ones)
#endif
// waves per SIMD the lazy launches are compiled for.  The item side of S-big is ~4 960 waves (two ranks per wave at
// d = 128): at four waves per SIMD (4 096 slots) the last fifth starts when the first waves retire, a second round.
#ifndef CREC_LAZY_ITEM_WAVES
#define CREC_LAZY_ITEM_WAVES 4 // 5 (96 VGPRs, 12 bytes of scratch) measured the same: 10.8 us
#endif
#define CREC_LAZY_WAVES(MODE) ((MODE) == 2 ? CREC_LAZY_ITEM_WAVES : 4)
#ifndef CREC_LAZY_ABL
#define CREC_LAZY_ABL 0 // timing ablations of bpr_triples_lazy (wrong results): 1 no finishers, 2 no publishing / tickets either,
#endif                  // 4 no LDS piece sums, 8 no optimizer arithmetic, 16 no loss terms / user descriptor
#ifndef CREC_IDX_RECS
#define CREC_IDX_RECS 12 // all of a thread's records at B = 6144: one round of gathers instead of three (a 20-batch call 227 -> 219 us)
#endif
#ifndef CREC_LAZY_HEADSTART
#define CREC_LAZY_HEADSTART 6 // x 0.43 us (tools/lazy_variants.sh)
#endif
#ifndef CREC_LAZY_DEEP
#define CREC_LAZY_DEEP 8 // contribution / partial rows in flight in a finisher's sum.  The d = 128 workgroup is 1024 threads, i.e. a
                         // hard limit of 128 VGPRs: 11 in flight spill three registers; 8 / 11 / 12 / 16 measured within 2 % of each other
                         // (tools/lazy_variants.sh)
#endif
#ifndef CREC_WIN_DEEP
#define CREC_WIN_DEEP 10 // the same in a lazy_window launch, for the front rows that keep their tickets (negative slots too); twelve spill in the
                         // 1024-thread geometry now that lazy_poll_front sits beside it
#endif
#ifndef CREC_FUSED_WT
#define CREC_FUSED_WT 0 // 1: the launch-per-step tf1 kernel writes its tables with write-through (sc1) stores
#endif

// ------------------------------------------------------------------------------------
// index workspace layout for one chunk of nbc batches
// ------------------------------------------------------------------------------------
struct IndexWs {
    int32_t *n_u, *n_q;        // [nbc] unique rows per batch
    int32_t *u_rows, *u_seg;   // [nbc][B], [nbc][B+1]   row id | heavy<<31 ; segment starts
    int4 *u_occ;               // [nbc][B]   {u, i, j, position in batch}
    int32_t *q_rows, *q_seg;   // [nbc][2B], [nbc][2B+1]
    int4 *q_occ;               // [nbc][2B]  {u, i, j, position | side<<31}  (side 1: row is the triple's j)
    int2 *u_rowseg, *q_rowseg; // tf1 mode: [nbc][U], [nbc][I]  {segment start, length (negative: heavy row)}
    int32_t *hn;               // [nbc][HN]  {user heavy rows, user chunks, item heavy rows, item chunks, user multi rows,
                               //             item multi rows, -, -}
    // lazy mode: per batch position and role (user, positive, negative) {row | single << 31, k, start, cnt}: the row
    // id (bit 31: it occurs once in the batch), the slot's position k in the side's sorted order, the row's segment
    // [start, start + cnt) there.  These are the lazy step's whole index: nothing else is written in lazy mode.
    int4 *u_desc, *i_desc, *j_desc; // [nbc][B]; u_desc by batch position, i_desc / j_desc by PROCESSING SLOT (see order)
    int4 *i_tmp, *j_tmp;            // [nbc][B] scratch of the index kernel (the item descriptors by batch position)
    int2 *slot_pu;                  // [nbc][B] by processing slot: {batch position, user row}: with i_desc / j_desc everything the
                                    // triple pass needs to address its rows comes from ONE round of coalesced loads
    // lazy mode: the order in which the triple pass takes the batch positions -- triples of heavy item rows first, then
    // those of other multi-occurrence item rows, then the rest: a multi-occurrence row is finished by its LAST
    // contributor, and a heavy row's finish is a chain of dependent round trips that should not start at the very end
    int32_t *order;                 // [nbc][B] batch position per processing slot
    int32_t *nslots;                // [nbc][2]  {triples of the batch, of which in the item-sorted front region}
    int32_t spw;                    // processing slots per workgroup of the triple pass
    // lazy mode, W consecutive batches in ONE launch (option lazy_window; bpr_triples_lazy<.., WIN>): per (batch, side) the sorted
    // unique rows, their segment starts and a presence filter -- what lazy_window_deps needs to find, for every row a triple
    // reads, the last EARLIER batch of the window that touches it -- and its result per processing slot
    int32_t win;                    // 0: off
    int32_t ids16;                  // 1: the batch's three id streams are kept in LDS as 16-bit ids (after the sort's counters): the records come from there
    int32_t *wn;                    // [nbc][2] unique rows of the side
    int32_t *wrow_u, *wst_u;        // [nbc][B]   row / segment start by rank
    int32_t *wrow_q, *wst_q;        // [nbc][2B]
    uint32_t *wfilt;                // [nbc][2][WF_WORDS] bit hash(row): the side names a row with this hash (zeroed per chunk)
    int4 *wdeps;                    // [nbc][B] by processing slot: {user, positive, negative, -}: -1 or batch-in-window << 20 | segment start
    int4 *u_hrows, *q_hrows;   // [nbc][HRu|HRq]  {row, first chunk, chunks<<16 | length, rank among unique rows}
    int4 *u_hitems, *q_hitems; // [nbc][HIu|HIq]  {row, segment start, length | heavy row index << 8, chunk index}
    int64_t B, U, I, HRu, HRq, HIu, HIq;
    int32_t CH;                // rows with more than CH occurrences are split into chunks of CH
    int32_t bits_u, bits_q;    // significant bits of a user / item row id (radix passes of the index sort)
    // tf1 mode, wide groups: the R consecutive table rows of one row wave that hold more than wide_total occurrences
    // together are handed to a whole workgroup (wide_group): its rows are flagged like heavy rows for the row wave
    int32_t wide_rows, wide_total; // R (0: off), threshold
    int32_t *u_wide, *q_wide;  // [nbc][WMu|WMq] group index = first row / R
    int32_t *nwide;            // [nbc][2] wide groups per batch {user, item}
    int64_t WMu, WMq;
    size_t bytes;
    // tf1 mode: the counts the host sizes the chunk's launches with reach it WITHOUT a copy and a stream synchronisation:
    // every workgroup of the index launch also writes its counts into page-locked, host-coherent memory (same layout as
    // [hn, nwide + 2 nbc)) and draws a ticket on `done`; the last one raises the flag the host is polling with the launch's
    // sequence number (null: the host copies)
    int32_t *host_counts, *done;
    int32_t *host_flag;
    int32_t seq, n_back;
    // bpr_steps_flow: rows are sorted (and u_rowseg / q_rowseg indexed) by POSITION instead of by id -- row r of a side sits in
    // workgroup r mod nwg, slot r div nwg, position (r mod nwg) * rpb + r div nwg (rpb rows per workgroup, a power of two) --
    // so U / I above are the position counts nwg * rpb.  nwg == 0: positions are row ids (every other step)
    int32_t flow_nwg_u, flow_nwg_q, flow_rpb_u, flow_rpb_q; // (rows per workgroup by side: item workgroups hold fewer, their rows are busier)
    uint32_t flow_magic_u, flow_magic_q; // floor(2^32 / nwg): the division by nwg as a multiplication + one fix-up
    // ... or by table (flow_deal_rows: rows dealt by how often they occur, so that the workgroups' record counts are even):
    // row -> position and position -> row (-1: nobody), per side; null: the arithmetic above
    const int32_t *flow_pos_u, *flow_pos_q, *flow_row_u, *flow_row_q;
};

// row id -> position under the row dealing of bpr_steps_flow
__device__ __forceinline__ uint32_t flow_pos_of(uint32_t row, uint32_t nwg, uint32_t magic, uint32_t rpb)
{
    uint32_t q = __umulhi(row, magic), rem = row - q * nwg;
    if (rem >= nwg) {
        ++q;
        rem -= nwg;
    }
    return rem * rpb + q;
}
__device__ __forceinline__ uint32_t flow_row_of(uint32_t pos, uint32_t nwg, uint32_t rpb) { return (pos & (rpb - 1)) * nwg + pos / rpb; }

constexpr int HN = 8; // per-batch counters the host reads back
constexpr int WF_WORDS = 4096, WF_SHIFT = 15; // lazy_window's presence filter: 2^17 bits per (batch, side)
__device__ __forceinline__ uint32_t wf_hash(uint32_t row) { return (row * 2654435761u) >> WF_SHIFT; }

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

inline int row_bits(int64_t n_rows)
{
    int b = 1;
    while (b < 32 && ((int64_t)1 << b) < n_rows)
        ++b;
    return b;
}

// processing slots per workgroup of bpr_triples_lazy: 32 lane groups; 64 at d = 4, where a group is one lane; 16 above
// d = 128, where a group is a whole wave (a workgroup has at most 1024 threads)
#ifndef CREC_LAZY_SPW
#define CREC_LAZY_SPW 16 // slots per workgroup at 4 < d <= 128: S-big kernel 19.5 / 17.7 / 18.6 us at 8 / 16 / 32 (tools/lazy_variants.sh)
#endif
// Batches of 4096 .. 8192 triples: 32 where the lane group is 8 .. 32 lanes (workgroups of up to 1024 threads) -- a popular item's
// positive slots then make half as many pieces, and its finisher sums them in one round of loads less: under lazy_window, where that
// finisher is on the path from batch to batch, S-big 13.0 -> 12.5 us per batch, ml-1m's catalogue at d = 128 20.9 -> 17.0 us per
// step; a launch per batch is as fast or faster so from 4096 triples (ml-1m d = 64: 14.1 -> 13.4), slower below (B = 2048: 11.9 ->
// 15.2).  The geometry depends on (d, B) only, never on the window: the pieces are part of the order the sums are formed in.
#ifndef CREC_LAZY_BIG_B
#define CREC_LAZY_BIG_B 4096
#endif
#ifndef CREC_LAZY_BIG_B_MAX
#define CREC_LAZY_BIG_B_MAX 8192 // (beyond the LDS-indexed batches the 512-thread geometry is the faster one: 58.0 against 63.5 us at 32 768, 108.3 / 111.7 at 65 536; tools/spw_bigbatch.sh)
#endif
inline int lazy_spw(int d, int64_t B = 0)
{
    if (d <= 4)
        return 64;
    if (d > 128)
        return 16;
    return (B >= CREC_LAZY_BIG_B && B <= CREC_LAZY_BIG_B_MAX && d > 16) ? 32 : (d > 64 && CREC_LAZY_SPW > 32 ? 32 : CREC_LAZY_SPW);
}

IndexWs layout_index(void *base, int64_t nbc, int64_t B, int64_t U, int64_t I, bool dense, int CH, int wide_rows = 0,
                     int wide_total = 0, int spw = 32, int flow_rpb = 0, int flow_nwg_u = 0, int flow_nwg_q = 0, int win = 0)
{
    IndexWs w{};
    w.B = B;
    if (dense && flow_rpb > 0) { // positions instead of row ids (IndexWs::flow_*); flow_rpb = rows per user workgroup | per item workgroup << 8
        w.flow_rpb_u = flow_rpb & 0xff;
        w.flow_rpb_q = flow_rpb >> 8;
        w.flow_nwg_u = flow_nwg_u > 0 ? flow_nwg_u : (int32_t)((U + w.flow_rpb_u - 1) / w.flow_rpb_u); // (the dealing by weight takes a few more)
        w.flow_nwg_q = flow_nwg_q > 0 ? flow_nwg_q : (int32_t)((I + w.flow_rpb_q - 1) / w.flow_rpb_q);
        w.flow_magic_u = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)w.flow_nwg_u > 0xffffffffull ? 0xffffffffull : ((uint64_t)1 << 32) / (uint64_t)w.flow_nwg_u);
        w.flow_magic_q = (uint32_t)(((uint64_t)1 << 32) / (uint64_t)w.flow_nwg_q > 0xffffffffull ? 0xffffffffull : ((uint64_t)1 << 32) / (uint64_t)w.flow_nwg_q);
        U = (int64_t)w.flow_nwg_u * w.flow_rpb_u;
        I = (int64_t)w.flow_nwg_q * w.flow_rpb_q;
    }
    w.U = dense ? U : 0;
    w.I = dense ? I : 0;
    w.CH = CH;
    w.bits_u = row_bits(U);
    w.bits_q = row_bits(I);
    w.HRu = dense ? B / (CH + 1) + 1 : 0;
    w.HRq = dense ? 2 * B / (CH + 1) + 1 : 0;
    w.HIu = dense ? B / CH + w.HRu + 1 : 0;
    w.HIq = dense ? 2 * B / CH + w.HRq + 1 : 0;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *p = base ? (char *)base + off : nullptr;
        off = align_up(off + bytes, 256);
        return (void *)p;
    };
    w.n_u = (int32_t *)take(sizeof(int32_t) * nbc * (dense ? 1 : 0));
    w.n_q = (int32_t *)take(sizeof(int32_t) * nbc * (dense ? 1 : 0));
    w.u_rows = (int32_t *)take(sizeof(int32_t) * nbc * (dense ? B : 0));
    w.u_seg = (int32_t *)take(sizeof(int32_t) * nbc * (dense ? B + 1 : 0));
    w.u_occ = (int4 *)take(sizeof(int4) * nbc * (dense ? B : 0));
    w.q_rows = (int32_t *)take(sizeof(int32_t) * nbc * (dense ? 2 * B : 0));
    w.q_seg = (int32_t *)take(sizeof(int32_t) * nbc * (dense ? 2 * B + 1 : 0));
    w.q_occ = (int4 *)take(sizeof(int4) * nbc * (dense ? 2 * B : 0));
    w.u_rowseg = (int2 *)take(sizeof(int2) * nbc * w.U);
    w.q_rowseg = (int2 *)take(sizeof(int2) * nbc * w.I);
    w.hn = (int32_t *)take(sizeof(int32_t) * nbc * (dense ? HN : 0));
    w.nwide = (int32_t *)take(sizeof(int32_t) * nbc * 2); // right behind hn: the host reads both back with one copy
    w.u_hrows = (int4 *)take(sizeof(int4) * nbc * w.HRu);
    w.q_hrows = (int4 *)take(sizeof(int4) * nbc * w.HRq);
    w.u_hitems = (int4 *)take(sizeof(int4) * nbc * w.HIu);
    w.q_hitems = (int4 *)take(sizeof(int4) * nbc * w.HIq);
    w.u_desc = (int4 *)take(sizeof(int4) * nbc * (dense ? 0 : B));
    w.i_desc = (int4 *)take(sizeof(int4) * nbc * (dense ? 0 : B));
    w.j_desc = (int4 *)take(sizeof(int4) * nbc * (dense ? 0 : B));
    w.order = (int32_t *)take(sizeof(int32_t) * nbc * (dense ? 0 : B));
    w.slot_pu = (int2 *)take(sizeof(int2) * nbc * (dense ? 0 : B));
    w.i_tmp = (int4 *)take(sizeof(int4) * nbc * (dense ? 0 : B));
    w.j_tmp = (int4 *)take(sizeof(int4) * nbc * (dense ? 0 : B));
    w.nslots = (int32_t *)take(sizeof(int32_t) * nbc * (dense ? 0 : 2));
    w.spw = spw;
    w.win = dense ? 0 : win;
    const int64_t wq = w.win ? 1 : 0;
    w.wn = (int32_t *)take(sizeof(int32_t) * nbc * 2 * wq);
    w.wrow_u = (int32_t *)take(sizeof(int32_t) * nbc * B * wq);
    w.wst_u = (int32_t *)take(sizeof(int32_t) * nbc * B * wq);
    w.wrow_q = (int32_t *)take(sizeof(int32_t) * nbc * 2 * B * wq);
    w.wst_q = (int32_t *)take(sizeof(int32_t) * nbc * 2 * B * wq);
    w.wfilt = (uint32_t *)take(sizeof(uint32_t) * nbc * 2 * WF_WORDS * wq);
    w.wdeps = (int4 *)take(sizeof(int4) * nbc * B * wq);
    w.wide_rows = dense && wide_total > 0 ? wide_rows : 0;
    w.wide_total = wide_total;
    w.WMu = w.wide_rows ? B / (wide_total + 1) + 1 : 0;
    w.WMq = w.wide_rows ? 2 * B / (wide_total + 1) + 1 : 0;
    w.u_wide = (int32_t *)take(sizeof(int32_t) * nbc * w.WMu);
    w.q_wide = (int32_t *)take(sizeof(int32_t) * nbc * w.WMq);
    w.bytes = off;
    return w;
}

// ------------------------------------------------------------------------------------
// bpr_index_batches: grid = 2*nbc workgroups; block b>>1 = batch, b&1 = side (0 user, 1 item).  tf1 mode, parts == 2: grid =
// 4*nbc -- every (batch, side) has TWO workgroups that both sort the side's pairs and then share the phases behind the sort:
// part 0 writes the heavy rows, the wide groups, the row segments and the counts, part 1 the occurrence records (they need the
// sorted arrays only).  What a call waits for is one workgroup's latency, and the CUs next door are idle: 46 -> 36 us.
// Stable radix sort in LDS of the side's (row, slot) pairs (block_sort.h): slots are the triples' positions on the
// user side, i-slices then j-slices on the item side, so that a row's occurrences come out in batch order, i before j.
// LDS: rows u32[cap], tag u16[cap], seg u16[cap+64], digit counters, scan scratch (cap = 2B rounded up to 64).
// ------------------------------------------------------------------------------------
inline size_t index_lds_bytes(int cap, int64_t ids16_B = 0)
{
    return sizeof(uint32_t) * cap + sizeof(uint16_t) * (2 * (size_t)cap + 64) +
           sizeof(int) * (crec_sort::hist_ints<PREP_THREADS>() + crec_sort::scan_ints<PREP_THREADS>()) + sizeof(uint16_t) * 3 * (size_t)ids16_B;
}

__global__ __launch_bounds__(PREP_THREADS) void bpr_index_batches(const int32_t *__restrict__ u,
                                                                  const int32_t *__restrict__ i,
                                                                  const int32_t *__restrict__ j, int64_t n_total,
                                                                  int32_t B, int32_t cap, IndexWs w, int32_t wg_parts)
{
    extern __shared__ uint32_t rows[]; // [cap]
    uint16_t *tag = (uint16_t *)(rows + cap);
    uint16_t *seg = tag + cap; // [cap + 64] run starts by rank, closed by n
    int *hist = (int *)(seg + cap + 64);
    int *scan = hist + crec_sort::hist_ints<PREP_THREADS>();
    uint16_t *ids = (uint16_t *)(scan + crec_sort::scan_ints<PREP_THREADS>()); // [3][B] when w.ids16
    const int pair = wg_parts == 2 ? (int)(blockIdx.x >> 1) : (int)blockIdx.x, part = wg_parts == 2 ? (int)(blockIdx.x & 1) : -1;
    const int batch = pair >> 1, side = pair & 1;
    const bool do_lists = part != 1, do_records = part != 0; // (wg_parts == 1: one workgroup does both)
    const int tid = threadIdx.x;
    const int64_t t0 = (int64_t)batch * B;
    const int nb = (int)((n_total - t0) < B ? (n_total - t0) : B); // triples in this batch
    const int n = side ? 2 * nb : nb;
    const int CH = w.CH;
    // The occurrence records name all three ids of a slot's triple: 2 x 12 288 x 3 scattered loads from the streams took the item
    // side's records workgroup 15.6 us of its 48 (one CU's address path); with catalogues below 65 536 rows the batch's streams are
    // read once, coalesced, into LDS as 16-bit ids and the records are put together from there
    if (w.ids16 && do_records)
        for (int k = tid; k < nb; k += PREP_THREADS) {
            ids[k] = (uint16_t)u[t0 + k];
            ids[B + k] = (uint16_t)i[t0 + k];
            ids[2 * B + k] = (uint16_t)j[t0 + k];
        }

    const uint32_t fl_nwg = (uint32_t)(side ? w.flow_nwg_q : w.flow_nwg_u), fl_magic = side ? w.flow_magic_q : w.flow_magic_u,
                   fl_rpb = (uint32_t)(side ? w.flow_rpb_q : w.flow_rpb_u);
    const int32_t *fl_pos = side ? w.flow_pos_q : w.flow_pos_u, *fl_row = side ? w.flow_row_q : w.flow_row_u;
    if (fl_pos) { // positions by table: a thread's ids first, then its look-ups, each round of loads issued together
        constexpr int FIF = 6;
        for (int k0f = tid; k0f < n; k0f += FIF * PREP_THREADS) {
            uint32_t r[FIF];
#pragma unroll
            for (int e = 0; e < FIF; ++e) {
                const int k = k0f + e * PREP_THREADS;
                r[e] = k < n ? (uint32_t)(side == 0 ? u[t0 + k] : (k < nb ? i[t0 + k] : j[t0 + k - nb])) : 0u;
            }
#pragma unroll
            for (int e = 0; e < FIF; ++e)
                r[e] = (uint32_t)fl_pos[r[e]];
#pragma unroll
            for (int e = 0; e < FIF; ++e) {
                const int k = k0f + e * PREP_THREADS;
                if (k < n) {
                    rows[k] = r[e];
                    tag[k] = (uint16_t)k;
                }
            }
        }
    } else {
        for (int k = tid; k < n; k += PREP_THREADS) {
            const uint32_t r = (uint32_t)(side == 0 ? u[t0 + k] : (k < nb ? i[t0 + k] : j[t0 + k - nb]));
            rows[k] = fl_nwg ? flow_pos_of(r, fl_nwg, fl_magic, fl_rpb) : r;
            tag[k] = (uint16_t)k;
        }
    }
#ifdef CREC_STAMPS // phase times of workgroups 0 .. 3 of the launch (batch 0: user / item side, or with two workgroups per side: user lists, user records, item lists, item records): crec_stamps[16000*8 + ...]
#define CREC_IDX_MARK(k)                                                           \
    do {                                                                           \
        __syncthreads();                                                           \
        if (tid == 0 && blockIdx.x < 4)                                            \
            crec_stamps[8 * (16000 + 4 * (k) + blockIdx.x)] = wall_clock64();      \
    } while (0)
#else
#define CREC_IDX_MARK(k)
#endif
    CREC_IDX_MARK(0);
    crec_sort::block_sort_rows<PREP_THREADS>(rows, tag, n, side ? w.bits_q : w.bits_u, hist, scan);
    CREC_IDX_MARK(1);

    // run starts (blocked over positions, so that the scan runs in sorted order) -> seg[rank]
    const int per = (n + PREP_THREADS - 1) / PREP_THREADS;
    const int k0 = tid * per, k1 = (k0 + per) < n ? (k0 + per) : n;
    int heads = 0;
    for (int k = k0; k < k1; ++k)
        heads += (k == 0 || rows[k] != rows[k - 1]) ? 1 : 0;
    int total = 0, hv_total = 0;
    int rank = crec_sort::block_excl_scan<PREP_THREADS>(heads, scan, &total);
    const int rank_first = rank; // runs that start before this thread's positions
    for (int k = k0; k < k1; ++k)
        if (k == 0 || rows[k] != rows[k - 1])
            seg[rank++] = (uint16_t)k;
    if (tid == 0)
        seg[total] = (uint16_t)n;
    __syncthreads();

    if ((side ? w.I : w.U) == 0) {
        // lazy mode.  The triple pass (bpr_triples_lazy) needs, per batch position and role, where the slot and its row
        // sit in the sorted order, and the order to take the positions in.  The triples whose POSITIVE item occurs more
        // than once as a positive come first, sorted by item: the positive slots of such a row are consecutive processing
        // slots, so the workgroups of the pass (SPW slots each) can add a row's contributions in LDS -- a row that lies
        // inside one workgroup is finished there without any memory round trip, a longer one leaves one partial per
        // workgroup it crosses instead of one contribution row per slot.
        if (w.win) { // (lazy_window) the side's unique rows by rank, their segment starts, the presence filter
            int32_t *wr = side ? w.wrow_q + (int64_t)batch * 2 * B : w.wrow_u + (int64_t)batch * B;
            int32_t *ws = side ? w.wst_q + (int64_t)batch * 2 * B : w.wst_u + (int64_t)batch * B;
            uint32_t *wf = w.wfilt + ((int64_t)batch * 2 + side) * WF_WORDS;
            for (int q = tid; q < total; q += PREP_THREADS) {
                const uint32_t r = rows[seg[q]];
                wr[q] = (int32_t)r;
                ws[q] = (int32_t)seg[q];
                const uint32_t hs = wf_hash(r);
                atomicOr(wf + (hs >> 5), 1u << (hs & 31u));
            }
            if (tid == 0)
                w.wn[batch * 2 + side] = total;
            __syncthreads(); // (the front rows' layout words below overwrite rows[start + 1])
        }
        uint16_t *wt = (uint16_t *)hist; // [B] per position: 0xffff = in the item-sorted front region, else the busier item row's count
        const int SPW = w.spw;
        int32_t *ord = w.order + (int64_t)batch * B;
        int n_front = 0; // triples in the front region
        if (side == 1) {
            const int rper = (total + PREP_THREADS - 1) / PREP_THREADS;
            const int q0 = tid * rper, q1 = (q0 + rper) < total ? (q0 + rper) : total;
            // a run's i-slots come first (stable sort, i-slices before j-slices): their number by binary search on the slot ids
            const auto i_slots = [&](int start, int cnt) {
                int lo = 0, hi = cnt;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if ((int)tag[start + mid] < nb)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
                return lo;
            };
            int mine = 0;
            for (int q = q0; q < q1; ++q) {
                const int start = seg[q], cnt = (int)seg[q + 1] - start;
                if (cnt >= 2) {
                    const int ci = i_slots(start, cnt);
                    mine += ci >= 2 ? ci : 0;
                }
            }
            int at = crec_sort::block_excl_scan<PREP_THREADS>(mine, scan, &n_front);
            for (int q = q0; q < q1; ++q) {
                const int start = seg[q], cnt = (int)seg[q + 1] - start;
                if (cnt >= 2) {
                    const int ci = i_slots(start, cnt);
                    if (ci >= 2) { // the run's row id stays in rows[start]; the next entry carries {positive slots, first processing slot}
                        rows[start + 1] = 0x80000000u | ((uint32_t)ci << 14) | (uint32_t)at;
                        at += ci;
                    }
                }
            }
            __syncthreads();
        }
        int ri = rank_first - 1; // index of the run that holds position k
        for (int k = k0; k < k1; ++k) {
            if (k == (int)seg[ri + 1])
                ++ri;
            const int start = seg[ri], cnt = (int)seg[ri + 1] - start;
            const uint32_t lay = (side == 1 && cnt >= 2) ? rows[start + 1] : 0u;
            const bool front = (lay & 0x80000000u) != 0;
            const int ci = front ? (int)((lay >> 14) & 0x3fffu) : 0, first = (int)(lay & 0x3fffu);
            const int slot = tag[k];
            // y: the slot's sorted position, and (front rows) where in its workgroup the row's first positive slot sits.
            // The item side's descriptors are written by batch position here and moved to processing-slot order below
            int4 *dst = side == 0 ? w.u_desc : (slot < nb ? w.i_desc : w.j_desc);
            int4 *dst_tmp = side == 0 ? dst : (slot < nb ? w.i_tmp : w.j_tmp);
            dst_tmp[(int64_t)batch * B + (slot < nb ? slot : slot - nb)] =
                lazy_desc_pack(rows[start], cnt == 1, k, start, cnt, ci, front ? first % SPW : 0);
            if (side == 1 && slot < nb) {
                if (front) { // (k - start < ci: a positive slot of a front row)
                    ord[first + (k - start)] = slot;
                    wt[slot] = 0xffff;
                } else {
                    wt[slot] = (uint16_t)(cnt > 0xfffe ? 0xfffe : cnt);
                }
            }
        }
        if (side == 0)
            return;
        __syncthreads();
        ri = rank_first - 1;
        for (int k = k0; k < k1; ++k) { // every position has one i-slot (above) and one j-slot (here): no two writers
            if (k == (int)seg[ri + 1])
                ++ri;
            const int cnt = (int)seg[ri + 1] - (int)seg[ri], slot = tag[k];
            if (slot >= nb && wt[slot - nb] != 0xffff && cnt > (int)wt[slot - nb])
                wt[slot - nb] = (uint16_t)(cnt > 0xfffe ? 0xfffe : cnt);
        }
        __syncthreads();
        // behind the front region: the positions with another multi-occurrence item row (finished by their LAST
        // contributor: the sooner they start the better), then the rest -- a stable partition
        const int pper = (nb + PREP_THREADS - 1) / PREP_THREADS;
        const int p0 = tid * pper, p1 = (p0 + pper) < nb ? (p0 + pper) : nb;
        int n_blk = 0, n_multi = 0;
        for (int q = p0; q < p1; ++q) {
            n_blk += wt[q] == 0xffff ? 1 : 0;
            n_multi += (wt[q] != 0xffff && wt[q] >= 2) ? 1 : 0;
        }
        int tot_pack = 0;
        const int before = crec_sort::block_excl_scan<PREP_THREADS>((n_blk << 16) | n_multi, scan, &tot_pack);
        const int tot_multi = tot_pack & 0xffff;
        int at_m = n_front + (before & 0xffff), at_s = n_front + tot_multi + (p0 - (before >> 16) - (before & 0xffff));
        for (int q = p0; q < p1; ++q) {
            if (wt[q] == 0xffff)
                continue;
            if (wt[q] >= 2)
                ord[at_m++] = q;
            else
                ord[at_s++] = q;
        }
        if (tid == 0) {
            w.nslots[batch * 2 + 0] = nb;
            w.nslots[batch * 2 + 1] = n_front;
        }
        // the order is complete (this workgroup wrote all of it): the item descriptors and {position, user} by processing slot
        __threadfence_block();
        __syncthreads();
        for (int sl = tid; sl < nb; sl += PREP_THREADS) {
            const int pos = ord[sl];
            w.slot_pu[(int64_t)batch * B + sl] = make_int2(pos, u[t0 + pos]);
            w.i_desc[(int64_t)batch * B + sl] = w.i_tmp[(int64_t)batch * B + pos];
            w.j_desc[(int64_t)batch * B + sl] = w.j_tmp[(int64_t)batch * B + pos];
        }
        return;
    }

    CREC_IDX_MARK(2);
    // heavy rows (more than CH occurrences) and their chunks, in rank order (blocked over ranks)
    int4 *hrows_out = side ? w.q_hrows + (int64_t)batch * w.HRq : w.u_hrows + (int64_t)batch * w.HRu;
    int4 *hitems_out = side ? w.q_hitems + (int64_t)batch * w.HIq : w.u_hitems + (int64_t)batch * w.HIu;
    const int rper = (total + PREP_THREADS - 1) / PREP_THREADS;
    const int q0 = tid * rper, q1 = (q0 + rper) < total ? (q0 + rper) : total;
    int hv = 0; // heavy rows << 16 | chunks
    for (int q = q0; q < q1; ++q) {
        const int cnt = (int)seg[q + 1] - (int)seg[q];
        if (cnt > CH)
            hv += (1 << 16) + (cnt + CH - 1) / CH;
    }
    const int hv_before = do_lists ? crec_sort::block_excl_scan<PREP_THREADS>(hv, scan, &hv_total) : 0;
    if (hv && do_lists) {
        int hrow = hv_before >> 16, hitem = hv_before & 0xffff;
        for (int q = q0; q < q1; ++q) {
            const int k = seg[q], cnt = (int)seg[q + 1] - k;
            if (cnt > CH) {
                const int row = (int)rows[k], parts = (cnt + CH - 1) / CH, my_h = hrow++;
                hrows_out[my_h] = make_int4(row, hitem, (parts << 16) | cnt, q);
                for (int p = 0; p < parts; ++p) {
                    const int len = (cnt - p * CH) < CH ? (cnt - p * CH) : CH;
                    hitems_out[hitem] = make_int4(row, k + p * CH, len | (my_h << 8), hitem);
                    ++hitem;
                }
            }
        }
    }

    CREC_IDX_MARK(3);
    // wide groups (tf1 mode): ranks whose rows share one row wave (row / R) and hold more than wide_total occurrences
    // together, none of them heavy.  The group's first present rank speaks for it.
    const int64_t n_rows = side ? w.I : w.U;
    const int WR = n_rows > 0 ? w.wide_rows : 0, WT_ = w.wide_total;
    // R is a power of two (lane groups per wave, capped at 4): a row's group is a shift away
    const int wshift = WR > 0 ? 31 - __builtin_clz((unsigned)WR) : 0;
    const auto group_of = [&](int q) { return rows[seg[q]] >> wshift; };
    const auto group_total = [&](int q) -> int { // q: first rank of its group; -1 if the group holds a heavy row
        const uint32_t gq = group_of(q);
        int tot = 0;
        for (int p = q; p < total && p < q + WR && group_of(p) == gq; ++p) {
            const int c = (int)seg[p + 1] - (int)seg[p];
            if (c > CH)
                return -1;
            tot += c;
        }
        return tot;
    };
    // one flag per rank (in the sort's counter space, free by now): the rank belongs to a wide group.  A group's first
    // present rank decides for the group
    uint8_t *wflag = (uint8_t *)hist;
    int wide_total_n = 0;
    if (WR > 0 && do_lists) {
        __syncthreads();
        for (int q = tid; q < total; q += PREP_THREADS) {
            if (q > 0 && group_of(q - 1) == group_of(q))
                continue;
            const bool wide = group_total(q) > WT_;
            const uint32_t gq = group_of(q);
            for (int p = q; p < total && p < q + WR && group_of(p) == gq; ++p)
                wflag[p] = wide ? 1 : 0;
        }
        __syncthreads();
        int nw = 0;
        for (int q = q0; q < q1; ++q)
            nw += (wflag[q] && (q == 0 || group_of(q - 1) != group_of(q))) ? 1 : 0;
        int at = crec_sort::block_excl_scan<PREP_THREADS>(nw, scan, &wide_total_n);
        int32_t *wide_out = side ? w.q_wide + (int64_t)batch * w.WMq : w.u_wide + (int64_t)batch * w.WMu;
        for (int q = q0; q < q1; ++q)
            if (wflag[q] && (q == 0 || group_of(q - 1) != group_of(q)))
                wide_out[at++] = (int32_t)group_of(q);
    }
    const auto in_wide = [&](int q) -> bool { return WR > 0 && wflag[q] != 0; };

    CREC_IDX_MARK(4);
    // per-rank outputs and (tf1 mode) every table row's {segment start, length; negative: heavy}, striped over ranks
    int32_t *rows_out = side ? w.q_rows + (int64_t)batch * 2 * B : w.u_rows + (int64_t)batch * B;
    int32_t *seg_out = side ? w.q_seg + (int64_t)batch * (2 * B + 1) : w.u_seg + (int64_t)batch * (B + 1);
    int2 *rs = side ? w.q_rowseg + (int64_t)batch * w.I : w.u_rowseg + (int64_t)batch * w.U;
    for (int q = tid; do_lists && q <= total; q += PREP_THREADS) {
        if (q < total) {
            const int k = seg[q], cnt = (int)seg[q + 1] - k;
            rows_out[q] = (int32_t)(rows[k] | (cnt > CH ? 0x80000000u : 0u));
            seg_out[q] = k;
        } else {
            seg_out[q] = n;
        }
        if (n_rows > 0)
            crec_sort::rowseg_fill(rs, n_rows, q, total, n, rows, seg, 0,
                                   [&](int c) { return (c > CH || in_wide(q)) ? -c : c; });
    }

    CREC_IDX_MARK(5);
    // occurrence records in sorted order
    int4 *occ_out = side ? w.q_occ + (int64_t)batch * 2 * B : w.u_occ + (int64_t)batch * B;
    constexpr int RIF = CREC_IDX_RECS; // records per thread in flight: the gathers are the cost (4: three rounds per item side)
    for (int k0r = tid; do_records && k0r < n; k0r += RIF * PREP_THREADS) {
        int pos[RIF];
        uint32_t payload[RIF];
        int4 rec[RIF];
#pragma unroll
        for (int e = 0; e < RIF; ++e) {
            const int k = k0r + e * PREP_THREADS;
            const int t = k < n ? tag[k] : 0;
            pos[e] = t < nb ? t : t - nb;
            payload[e] = (t < nb ? 0u : 0x80000000u) | (uint32_t)pos[e]; // side flag: the row is the triple's j
            if (fl_nwg && k < n) // bpr_steps_flow: bits 27 .. 30 carry the slot of the record's own row in its workgroup
                payload[e] |= (rows[k] & (fl_rpb - 1u)) << 27;
        }
#pragma unroll
        for (int e = 0; e < RIF; ++e) { // the sorted row is one of the three ids: two scattered loads per record, not three
            const int k = k0r + e * PREP_THREADS;
            if (w.ids16) {
                rec[e] = make_int4((int)ids[pos[e]], (int)ids[B + pos[e]], (int)ids[2 * B + pos[e]], (int)payload[e]);
                continue;
            }
            if (fl_row) { // (positions by table: the sorted key is no row id; all three ids come from the stream, independent loads)
                rec[e] = make_int4(u[t0 + pos[e]], i[t0 + pos[e]], j[t0 + pos[e]], (int)payload[e]);
                continue;
            }
            const int row = k < n ? (int)(fl_nwg ? flow_row_of(rows[k], fl_nwg, fl_rpb) : rows[k]) : 0;
            if (side == 0)
                rec[e] = make_int4(row, i[t0 + pos[e]], j[t0 + pos[e]], (int)payload[e]);
            else if ((int)payload[e] < 0)
                rec[e] = make_int4(u[t0 + pos[e]], i[t0 + pos[e]], row, (int)payload[e]);
            else
                rec[e] = make_int4(u[t0 + pos[e]], row, j[t0 + pos[e]], (int)payload[e]);
        }
#pragma unroll
        for (int e = 0; e < RIF; ++e)
            if (k0r + e * PREP_THREADS < n)
                occ_out[k0r + e * PREP_THREADS] = rec[e];
    }
    CREC_IDX_MARK(6);
    if (!do_lists)
        return;
    if (tid == 0) {
        (side ? w.n_q : w.n_u)[batch] = total;
        w.hn[batch * HN + side * 2 + 0] = hv_total >> 16;
        w.hn[batch * HN + side * 2 + 1] = hv_total & 0xffff;
        w.nwide[batch * 2 + side] = wide_total_n;
    }
    if (w.host_counts && tid == 0) { // the chunk's counts straight into the host's hands (see IndexWs)
        // No fences (a device- or system-scope fence writes the L2 back: 10+ us at the end of a kernel that has just written
        // megabytes of records): system-scope stores travel by themselves, the wave waits for their completion, then draws
        // its ticket; the workgroup that draws the last one raises the flag.
        int32_t *hc = w.host_counts + batch * HN + side * 2, *hw = w.host_counts + (w.nwide - w.hn) + batch * 2 + side;
        __hip_atomic_store(hc + 0, (int32_t)(hv_total >> 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(hc + 1, (int32_t)(hv_total & 0xffff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(hw, (int32_t)wide_total_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // slots 6, 7 of a batch's counters are free: this workgroup's stamp, which the host checks before it trusts the rest
        __hip_atomic_store(w.host_counts + batch * HN + 6 + side, w.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (__hip_atomic_fetch_add(w.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x / (wg_parts == 2 ? 2 : 1) - 1) {
            __hip_atomic_store(w.done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch
            __hip_atomic_store(w.host_flag, w.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------------------------
// shared device pieces
// ------------------------------------------------------------------------------------
struct Hyper {
    float lr_t, beta1, beta2, eps, reg;
    int opt; // 0 Adam; CREC_OPT_SGD, CREC_OPT_ADAGRAD: the touched-rows-only optimizers of utils/tools.py:79-87
    float *dQ; // data-parallel step: item rows' gradient sums go here ([I,d], every row written) instead of Adam
    int dq_sys = 0; // dQ is a peer-transport window (< 2 GiB) that other GPUs will read: system-scope (written-through) stores, see st4_dq
    // deferred mode (exact TF1 Adam with the traffic of the lazy step): TF1 decays m, v and moves var on EVERY row at
    // every step; here a row carries the step at which its (var, m, v) were last brought up to date and replays the
    // steps it skipped -- m *= b1, v *= b2, var -= lr_s*m/(sqrt(v)+eps) for s = last+1 .. -- whenever it is touched
    // (before its values are used) and before evaluation (crec_bpr_adam_flush)
    const float *lr_tab;     // lr_s by step s (1-based), device
    int32_t *lastP, *lastQ;  // per row: the step its state is current at
    int32_t step;            // the step being applied
    // lazy_window: where whoever finishes a row leaves this batch's stamp (by segment start; null: a launch per batch)
    int32_t *win_done_u = nullptr, *win_done_q = nullptr;
    int32_t win_stamp = 0;
    int win_wb = 0; // (the batch's place in the launch: debug timeline)
    bool win_fit_p = false, win_fit_q = false; // the user / item tables lie within the 4 GiB one buffer descriptor spans
};

struct Tables { // one buffer set
    float *P, *mP, *vP, *Q, *mQ, *vQ;
};

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
// bit 0: the rows a step writes (var, m, v) go out with non-temporal stores: nothing reads them again before the
// launch ends, and lines that do not stay dirty in the L2s shorten what the end of the kernel has to write back
// (7.1 -> 6.4 us per launch at ml-1m shape).  bit 1: non-temporal loads of m and v (no gain, off).
#ifndef CREC_NT
#define CREC_NT 1
#endif
#ifndef CREC_TICKET_FENCES
#define CREC_TICKET_FENCES 0 // 1: publish heavy-row partials with plain stores + release/acquire fences
#endif
// device-coherent float4 traffic for the heavy-row partials: agent-scope (sc1) accesses are served at the level the
// XCDs share, so publishing a partial needs no L2 write-back and reading it no L2 invalidate
__device__ __forceinline__ void st4_agent(float *p, float4 v)
{
    __hip_atomic_store(p + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(p + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 ld4_agent(const float *p)
{
    return make_float4(__hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                       __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
typedef float crec_f4v __attribute__((ext_vector_type(4)));
// streamed table output: rows written once per launch and not re-read by it
__device__ __forceinline__ void st4s(float *p, float4 v)
{
    if (CREC_NT & 1) {
        crec_f4v t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<crec_f4v *>(p));
    } else {
        st4(p, v);
    }
}
// write-through (agent-scope, sc1) 16-byte store: the line does not stay dirty in this XCD's L2, so a later device-wide
// barrier needs no L2 write-back to publish it.  base: wave-uniform table pointer; off: float index (< 2^29)
__device__ __forceinline__ void st4_wt(float *base, int64_t off, float4 v)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    const crec_u4v x = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(x, r, (int)(off * 4), 0, 16 /* sc1 */);
}
// A gradient row for an exchange: into a plain buffer, or -- sys -- into this rank's peer-transport window with a SYSTEM-scope
// (sc0 sc1, written through) store, the form crec_peer.hip's own copy-in uses and its canary proves between the node's GPUs: what
// the peers read does not depend on which scope the end of this kernel releases at.  base: wave-uniform; off: float index
__device__ __forceinline__ void st4_dq(float *base, int64_t off, float4 v, bool sys, bool stream)
{
    if (sys) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
        const crec_u4v x = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
        __builtin_amdgcn_raw_buffer_store_b128(x, r, (int)(off * 4), 0, 17 /* sc0 sc1 */);
    } else if (stream) {
        st4s(base + off, v);
    } else {
        st4(base + off, v);
    }
}
// the matching 16-byte load: agent scope (sc1), served where the XCDs meet -- one instruction instead of the four 4-byte
// atomic loads of ld4_agent.  base: wave-uniform; off: float index (< 2^29)
__device__ __forceinline__ float4 ld4_sc1(const float *base, int64_t off)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
    const crec_u4v x = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off * 4), 0, 16 /* sc1 */);
    return make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w));
}
// the same load without a scope bit (cached like a plain load; a scalar base and a 32-bit offset instead of a 64-bit address per lane)
__device__ __forceinline__ float4 ld4_buf(const float *base, int64_t off)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
    const crec_u4v x = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off * 4), 0, 0);
    return make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w));
}
// lazy_window's table accesses: 16-byte agent-scope loads / written-through stores at full 64-bit addresses (the user tables of
// S-big are 5 GB each: beyond what one buffer descriptor spans, and a descriptor per lane costs a waterfall loop per load).  The
// compiler has no builtin for a global load with the sc1 bit, and it does not track what inline assembly leaves in flight -- a
// destination register of an assembly load may be copied or reused before the data is there (seen: garbage rows) -- so a round
// of loads AND its s_waitcnt are ONE assembly statement: nothing is outstanding where the compiler takes over again.
// win_load9: nine rows' slices at once (the three rows of a triple and their slots), one wait.
__device__ __forceinline__ void win_load9(const float *const (&a)[9], float4 (&r)[9])
{
    crec_f4v x0, x1, x2, x3, x4, x5, x6, x7, x8;
    asm volatile("global_load_dwordx4 %0, %9, off sc1\n\t"
                 "global_load_dwordx4 %1, %10, off sc1\n\t"
                 "global_load_dwordx4 %2, %11, off sc1\n\t"
                 "global_load_dwordx4 %3, %12, off sc1\n\t"
                 "global_load_dwordx4 %4, %13, off sc1\n\t"
                 "global_load_dwordx4 %5, %14, off sc1\n\t"
                 "global_load_dwordx4 %6, %15, off sc1\n\t"
                 "global_load_dwordx4 %7, %16, off sc1\n\t"
                 "global_load_dwordx4 %8, %17, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7), "=&v"(x8)
                 : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8])
                 : "memory");
    r[0] = make_float4(x0.x, x0.y, x0.z, x0.w), r[1] = make_float4(x1.x, x1.y, x1.z, x1.w), r[2] = make_float4(x2.x, x2.y, x2.z, x2.w);
    r[3] = make_float4(x3.x, x3.y, x3.z, x3.w), r[4] = make_float4(x4.x, x4.y, x4.z, x4.w), r[5] = make_float4(x5.x, x5.y, x5.z, x5.w);
    r[6] = make_float4(x6.x, x6.y, x6.z, x6.w), r[7] = make_float4(x7.x, x7.y, x7.z, x7.w), r[8] = make_float4(x8.x, x8.y, x8.z, x8.w);
}
// a finisher's two slots (m, v): asked for and waited for in one statement as well
__device__ __forceinline__ void win_load2(const float *a0, const float *a1, float4 &r0, float4 &r1)
{
    crec_f4v x0, x1;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                 "global_load_dwordx4 %1, %3, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(x0), "=&v"(x1)
                 : "v"(a0), "v"(a1)
                 : "memory");
    r0 = make_float4(x0.x, x0.y, x0.z, x0.w), r1 = make_float4(x1.x, x1.y, x1.z, x1.w);
}
// one row's slice and its two slots
__device__ __forceinline__ void win_load3(const float *a0, const float *a1, const float *a2, float4 &r0, float4 &r1, float4 &r2)
{
    crec_f4v x0, x1, x2;
    asm volatile("global_load_dwordx4 %0, %3, off sc1\n\t"
                 "global_load_dwordx4 %1, %4, off sc1\n\t"
                 "global_load_dwordx4 %2, %5, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(x0), "=&v"(x1), "=&v"(x2)
                 : "v"(a0), "v"(a1), "v"(a2)
                 : "memory");
    r0 = make_float4(x0.x, x0.y, x0.z, x0.w), r1 = make_float4(x1.x, x1.y, x1.z, x1.w), r2 = make_float4(x2.x, x2.y, x2.z, x2.w);
}
// written-through store (the caller's s_waitcnt vmcnt(0) in front of its stamp waits for the acknowledgement).  s_nop 1: on
// gfx940 and later a store of more than 8 bytes needs TWO wait states before a VALU instruction may write its data registers
// again, and the compiler's hazard bookkeeping does not look inside an assembly statement (seen in round 4 with the same
// statement in tcf_triples: store data lost to the next row's arithmetic; crec_store.h)
__device__ __forceinline__ void st4_wtg(float *p, float4 v)
{
    const crec_f4v x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
}
// the lazy step's published rows (contributions, partials): agent-scope accesses + s_waitcnt, or -- CREC_TICKET_FENCES,
// the textbook form kept buildable and compared bit for bit under test (tests/test_gpu_fences.py) -- plain accesses
// bracketed by device-scope fences
__device__ __forceinline__ void st4_pub(float *base, int64_t off, float4 v)
{
    if (CREC_TICKET_FENCES)
        st4(base + off, v);
    else
        st4_wt(base, off, v);
}
__device__ __forceinline__ float4 ld4_pub(const float *base, int64_t off) { return CREC_TICKET_FENCES ? ld4(base + off) : ld4_sc1(base, off); }
__device__ __forceinline__ void pub_release()
{
    if (CREC_TICKET_FENCES)
        __threadfence();
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void pub_acquire()
{
    if (CREC_TICKET_FENCES)
        __threadfence();
    else
        asm volatile("" ::: "memory");
}
__device__ __forceinline__ float4 ld4s(const float *p) // streamed slot input (m, v: read once)
{
    if (CREC_NT & 2) {
        const crec_f4v t = __builtin_nontemporal_load(reinterpret_cast<const crec_f4v *>(p));
        return make_float4(t.x, t.y, t.z, t.w);
    }
    return ld4(p);
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }

template <int CTRL> __device__ __forceinline__ float dpp_add(float v)
{
    // v + (v read through the DPP lane pattern CTRL); all lanes active
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// all-reduce over the LPR lanes of a group; every lane ends with the bit-identical sum.
// xor-1 / xor-2 inside quads, then mirror patterns (a quad, then an 8-lane half, already holds one
// value, so a mirror is as good as an xor), then real cross-row shuffles above 16 lanes.
template <int LPR> __device__ __forceinline__ float group_sum(float v)
{
    if (LPR >= 2)
        v = dpp_add<0xB1>(v); // quad_perm [1,0,3,2]
    if (LPR >= 4)
        v = dpp_add<0x4E>(v); // quad_perm [2,3,0,1]
    if (LPR >= 8)
        v = dpp_add<0x141>(v); // row_half_mirror
    if (LPR >= 16)
        v = dpp_add<0x140>(v); // row_mirror
    if (LPR >= 32)
        v += __shfl_xor(v, 16, 64);
    if (LPR >= 64)
        v += __shfl_xor(v, 32, 64);
    return v;
}

// sum over the G groups of a wave (same lane position in each group)
template <int LPR> __device__ __forceinline__ float4 cross_group_sum(float4 g)
{
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        if (off == 1) {
            g.x = dpp_add<0xB1>(g.x), g.y = dpp_add<0xB1>(g.y), g.z = dpp_add<0xB1>(g.z), g.w = dpp_add<0xB1>(g.w);
        } else if (off == 2) {
            g.x = dpp_add<0x4E>(g.x), g.y = dpp_add<0x4E>(g.y), g.z = dpp_add<0x4E>(g.z), g.w = dpp_add<0x4E>(g.w);
        } else {
            g.x += __shfl_xor(g.x, off, 64);
            g.y += __shfl_xor(g.y, off, 64);
            g.z += __shfl_xor(g.z, off, 64);
            g.w += __shfl_xor(g.w, off, 64);
        }
    }
    return g;
}

// Eigen softplus as TF1 evaluates -log_sigmoid (utils/tools.py:70-71)
__device__ __forceinline__ float tf_softplus(float z)
{
    const float thr = -13.942385f; // log(FLT_EPSILON) + 2
    if (z > -thr)
        return z;
    if (z < thr)
        return expf(z);
    return log1pf(expf(z));
}

// The two quotients of the step.  An IEEE-exact fp32 divide or square root is a ~15-instruction software
// sequence on gfx950 and these kernels are instruction-issue bound, so by default both use
// the hardware v_rcp_f32 / v_sqrt_f32 / v_exp_f32 (<= 1 ulp each): the update differs from the exactly
// rounded one by <= ~3e-7 of ITSELF (an update is ~lr = 1e-3), five orders below the 1e-5 parity bar.
// -DCREC_EXACT_MATH restores the exactly rounded forms (bit-reproducible against the CPU oracle's Adam).
#ifdef CREC_EXACT_MATH
__device__ __forceinline__ float adam_quot(float num, float v, float eps) { return num / (sqrtf(v) + eps); }
__device__ __forceinline__ float sigmoid_neg(float x) { return 1.0f / (1.0f + expf(x)); }
#else
__device__ __forceinline__ float adam_quot(float num, float v, float eps)
{
    return num * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + eps);
}
__device__ __forceinline__ float sigmoid_neg(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(x)); }
#endif

// TF1 Adam on one float4 of a row.  touched == false: the g = 0 sweep of a row the batch does not
// index (m, v still decay and var still moves: _apply_sparse_shared).
__device__ __forceinline__ void adam4(float4 &var, float4 &m, float4 &v, float4 g, bool touched, const Hyper &h)
{
    const float omb1 = 1.0f - h.beta1, omb2 = 1.0f - h.beta2;
    float *pv = &var.x, *pm = &m.x, *pw = &v.x;
    const float *pg = &g.x;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float mk = pm[c] * h.beta1;
        float vk = pw[c] * h.beta2;
        if (touched) {
            mk = mk + pg[c] * omb1;
            vk = vk + (pg[c] * pg[c]) * omb2;
        }
        pm[c] = mk;
        pw[c] = vk;
        pv[c] = pv[c] - adam_quot(h.lr_t * mk, vk, h.eps);
    }
}

// The update of a touched row under the configured optimizer (utils/tools.py:79-87), gradient already summed over the
// row's occurrences (TF1 deduplicates IndexedSlices before _apply_sparse).
//   Adam     see adam4 (lazy form: touched rows only)
//   SGD      GradientDescentOptimizer: var -= lr * g
//   Adagrad  AdagradOptimizer(lr), initial_accumulator_value 0.1 (the caller initialises the slot `m` to it):
//            accum += g*g; var -= lr * g / sqrt(accum)
__device__ __forceinline__ void opt4(float4 &var, float4 &m, float4 &v, float4 g, const Hyper &h)
{
    if (CREC_LAZY_ABL & 8) { // timing ablation: no optimizer arithmetic
        var = make_float4(var.x + g.x, var.y + g.y, var.z + g.z, var.w + g.w);
        m = make_float4(m.x + g.x, m.y + g.y, m.z + g.z, m.w + g.w);
        v = make_float4(v.x + g.x, v.y + g.y, v.z + g.z, v.w + g.w);
        return;
    }
    if (h.opt == 0) {
        adam4(var, m, v, g, true, h);
        return;
    }
    float *pv = &var.x, *pm = &m.x;
    const float *pg = &g.x;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (h.opt == CREC_OPT_SGD) {
            pv[c] = pv[c] - h.lr_t * pg[c];
        } else {
            const float a = pm[c] + pg[c] * pg[c];
            pm[c] = a;
#ifdef CREC_EXACT_MATH
            pv[c] = pv[c] - (h.lr_t * pg[c]) / sqrtf(a);
#else
            pv[c] = pv[c] - (h.lr_t * pg[c]) * __builtin_amdgcn_rsqf(a);
#endif
        }
    }
}

// deferred mode: bring a row's float4 slice from the state after step `last` to the state after step `upto` with the
// untouched-row update of TF1's sparse Adam.  m falls below the smallest fp32 denormal after < 1024 steps (0.9^1024 =
// 1e-47 of itself; the entry point refuses beta1 > 0.9), from where on var no longer moves: the rest of the gap is
// v *= b2^n -- one powf instead of n fp32 multiplications, i.e. v is within a few ulp of the sweep's, not bit-identical, for
// rows idle longer than 1024 steps (and m, v likewise from the step at which var has stopped moving: adam_replay).
constexpr int CREC_REPLAY_MAX = 1024;
__device__ __forceinline__ void adam_replay(float4 &var, float4 &m, float4 &v, int last, int upto, const Hyper &h)
{
    const int n = upto - last;
    if (n <= 0 || last == 0) // a row that was never updated has m = v = 0: the replayed update is var -= 0
        return;
    const int nr = n < CREC_REPLAY_MAX ? n : CREC_REPLAY_MAX;
    Hyper hs = h;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    // var -= lr_s m_s / (sqrt(v_s) + eps) shrinks by 0.9 / sqrt(0.999) per idle step (lr_s grows by less than 0.1 % per step): once a
    // step leaves every lane's var as it was, no later one moves it either, and only m *= b1, v *= b2 remain -- the same two
    // multiplications adam4 does, taken in closed form.  Rows idle for ~700 steps (the steady state of a 4 M x 400 k table;
    // tools/time_deferred_replay.py): 571 us per step with every step replayed in full, 347 with the multiplications one by one, 264 in
    // closed form (under lazy_window 354 / 185 / 142); runs shorter than ~250 steps pay 10-25 % for the test.  The exit is the wave's:
    // lanes of other rows may still be moving.
    int k = 0;
    for (; k < nr; ++k) {
        hs.lr_t = h.lr_tab[last + 1 + k];
        const float4 before = var;
        adam4(var, m, v, zero, false, hs);
        const bool moved = var.x != before.x || var.y != before.y || var.z != before.z || var.w != before.w;
        if (__builtin_amdgcn_ballot_w64(moved) == 0) {
            ++k;
            break;
        }
    }
    if (k < nr) { // the rest of the gap in closed form: m, v within a few ulp of the multiplications one by one (as beyond CREC_REPLAY_MAX below)
        const float fm = powf(h.beta1, (float)(nr - k)), fv = powf(h.beta2, (float)(nr - k));
        m = make_float4(m.x * fm, m.y * fm, m.z * fm, m.w * fm);
        v = make_float4(v.x * fv, v.y * fv, v.z * fv, v.w * fv);
    }
    if (n > nr) {
        const float f = powf(h.beta2, (float)(n - nr));
        v = make_float4(v.x * f, v.y * f, v.z * f, v.w * f);
        m = zero;
    }
}

__device__ __forceinline__ int4 shfl4(int4 v, int src)
{
    return make_int4(__shfl(v.x, src, 64), __shfl(v.y, src, 64), __shfl(v.z, src, 64), __shfl(v.w, src, 64));
}
__device__ __forceinline__ float4 shfl4f(float4 v, int src)
{
    return make_float4(__shfl(v.x, src, 64), __shfl(v.y, src, 64), __shfl(v.z, src, 64), __shfl(v.w, src, 64));
}
__device__ __forceinline__ float4 shfl4f_xor(float4 v, int mask)
{
    return make_float4(__shfl_xor(v.x, mask, 64), __shfl_xor(v.y, mask, 64), __shfl_xor(v.z, mask, 64),
                       __shfl_xor(v.w, mask, 64));
}
__device__ __forceinline__ void add4(float4 &a, float4 b)
{
    a.x = a.x + b.x;
    a.y = a.y + b.y;
    a.z = a.z + b.z;
    a.w = a.w + b.w;
}

// x_b and s_b of one triple from the group's float4 slices: the ONE place this arithmetic lives,
// so user-row and item-row waves agree bit for bit.
// `part` is the lane's share of p.q_i - p.q_j (its float4 slice of both products, subtracted before the cross-lane
// sum: one reduction instead of two).  Item-row waves hold the row they own as q_i or as q_j and form the same share
// from dot4(p, own) and dot4(p, other) with an exact sign flip, so the bits agree with the user-row waves'.
template <int LPR> __device__ __forceinline__ float triple_s(float part, float *x_out)
{
    if (CREC_ABL & 2) {
        *x_out = part;
        return part;
    }
    const float x = group_sum<LPR>(part); // ui_scores - uj_scores (BPR.py:39-42)
    *x_out = x;
    return sigmoid_neg(x); // sigmoid(-x) = -d softplus(-x)/dx
}


// ------------------------------------------------------------------------------------
// A normal wave owns R = min(G, ROWS_PER_WAVE) consecutive table rows ("slots"), one per lane
// group: group g holds slot g's row, m and v slices as float4s (every row byte is loaded and stored
// exactly once, all lanes busy in the Adam arithmetic).  The occurrences of consecutive rows are
// consecutive in the sorted occurrence list, so the wave walks contiguous ranges of records, G at
// a time: the group that draws occurrence k finds its slot from the slots' segment ends, borrows
// that slot's row slice with a cross-lane read, gathers the other rows, and adds the result into a
// per-slot accumulator.  A reduce-scatter over the groups then hands slot g's gradient to group g.
//
// Heavy rows (more than CH occurrences in the batch: popular items) would make their wave the
// kernel's critical path, so the index kernel cuts them into chunks of CH: chunk waves (R = 1, all
// G groups on one row) write partial gradients, normal waves skip such a slot, and the chunk wave that
// draws the row's last ticket adds the partials in chunk order and applies Adam.  Every sum keeps one fixed order.
// ------------------------------------------------------------------------------------
// RW = rows per wave requested | (FULL << 8); FULL: d == 4 * LPR, every lane owns a live float4 (lets the
// compiler drop the per-lane "inside the row" predicate and its exec-mask branches)
template <int LPR, int RW> struct Geo {
    static constexpr int G = 64 / LPR;
    static constexpr int R = G < (RW & 0xff) ? G : (RW & 0xff);
    static constexpr bool FULL = ((RW >> 8) & 1) != 0;
};

template <int LPR, int RW> using AccArr = float4[Geo<LPR, RW>::R];

struct Slot { // per lane: the slot of this lane's group (meaningful when g < n_valid)
    int64_t row;
    int start, cnt; // cnt < 0: heavy row, handled by chunk waves (chunk_wave)
};

// sum over groups of acc[s], delivered to group s (slot owner); other groups' return value is unused
template <int LPR, int RW> __device__ __forceinline__ float4 reduce_to_owner(AccArr<LPR, RW> &acc)
{
    constexpr int G = Geo<LPR, RW>::G, R = Geo<LPR, RW>::R;
    const int g = (threadIdx.x & 63) / LPR;
    if (G == 1)
        return acc[0];
    if (R == G && G == 2) { // partner = other group: keep my slot, add the partner's partial of it
        const float4 give = g == 0 ? acc[R - 1] : acc[0];
        float4 mine = g == 0 ? acc[0] : acc[R - 1];
        add4(mine, shfl4f_xor(give, LPR));
        return mine;
    }
    if (R == G && G == 4) {
        // step 1, partner g^2: keep the slot pair {g&2, (g&2)+1}, hand over the other pair
        const bool hi = (g & 2) != 0;
        float4 k0 = hi ? acc[2 % R] : acc[0], k1 = hi ? acc[3 % R] : acc[1 % R];
        const float4 s0 = hi ? acc[0] : acc[2 % R], s1 = hi ? acc[1 % R] : acc[3 % R];
        add4(k0, shfl4f_xor(s0, 2 * LPR));
        add4(k1, shfl4f_xor(s1, 2 * LPR));
        // step 2, partner g^1: keep slot g, hand over the pair's other slot
        const bool odd = (g & 1) != 0;
        float4 mine = odd ? k1 : k0;
        const float4 give = odd ? k0 : k1;
        add4(mine, shfl4f_xor(give, LPR));
        return mine;
    }
    // general: all-reduce every slot over all groups, then pick
    float4 out = acc[0];
#pragma unroll
    for (int s = 0; s < R; ++s) {
        float4 a = acc[s];
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1)
            add4(a, shfl4f_xor(a, off));
        if (g == s)
            out = a;
    }
    return out;
}

// MODE 0: user rows (gathers q_i, q_j; stores x_b; lazy mode also stores c_b = s_b * p_u)
// MODE 1: item rows, fused step (gathers p_u and the triple's other item row; recomputes s_b)
// MODE 2: item rows, lazy mode (reads c_b back)
// MODE 3: user rows, lazy mode (reads the stored s_b*(q_i-q_j) back) -- chunk waves of heavy rows only
//
// One "trip" = UNR sub-iterations of G occurrences.  Trips are software-pipelined two deep: the
// gathers of trip n+1 are issued before trip n is consumed.
struct Trip {
    int4 rec[UNR];
    float4 a[UNR], b[UNR];
    bool has[UNR];
    int slot[UNR];
};

template <int R> struct WaveCtx {
    const float *P, *Q;
    float *contrib, *xb;
    int d, g, l;
    bool act;
    int e[R]; // wave-uniform segment ends of the slots
    float reg;
};

template <int LPR, int RW, int MODE>
__device__ __forceinline__ void trip_issue(Trip &tr, const WaveCtx<Geo<LPR, RW>::R> &w, const int4 &mine, int c0, int nc,
                                           int n_sub, int t0)
{
    constexpr int G = Geo<LPR, RW>::G, R = Geo<LPR, RW>::R;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < UNR; ++t) {
        const int kk = (t0 + t) * G + w.g;
        tr.rec[t] = shfl4(mine, kk & 63);
        tr.has[t] = (t0 + t) < n_sub && kk < nc;
        const int k = c0 + kk;
        int s = 0;
#pragma unroll
        for (int q = 0; q + 1 < R; ++q)
            s += k >= w.e[q];
        tr.slot[t] = s;
        tr.a[t] = tr.b[t] = zero;
        // FULL: a dead slot's record is all zeros -> row 0, a harmless load, so no exec-mask branch is needed
        if (Geo<LPR, RW>::FULL || (tr.has[t] && w.act)) {
            if (CREC_ABL & 1) {
                tr.a[t] = ld4(w.Q + (int64_t)(tr.rec[t].y & 3) * w.d + 4 * w.l);
                tr.b[t] = ld4(w.Q + (int64_t)(tr.rec[t].z & 3) * w.d + 4 * w.l);
            } else if (MODE == 0) {
                tr.a[t] = ld4(w.Q + (int64_t)tr.rec[t].y * w.d + 4 * w.l);
                tr.b[t] = ld4(w.Q + (int64_t)tr.rec[t].z * w.d + 4 * w.l);
            } else if (MODE == 1) {
                tr.a[t] = ld4(w.P + (int64_t)tr.rec[t].x * w.d + 4 * w.l);
                const int other = tr.rec[t].w < 0 ? tr.rec[t].y : tr.rec[t].z;
                tr.b[t] = ld4(w.Q + (int64_t)other * w.d + 4 * w.l);
            } else {
                tr.a[t] = ld4(w.contrib + (int64_t)(tr.rec[t].w & 0x7fffffff) * w.d + 4 * w.l);
            }
        }
    }
}

template <int LPR, int RW, int MODE, bool STORE_CONTRIB>
__device__ __forceinline__ void trip_consume(const Trip &tr, const WaveCtx<Geo<LPR, RW>::R> &w, float4 var,
                                             AccArr<LPR, RW> &acc)
{
    constexpr int R = Geo<LPR, RW>::R;
#pragma unroll
    for (int t = 0; t < UNR; ++t) {
        // dead sub-iterations (past the end of the range) have has == false in every lane
        const float4 own = R == 1 ? var : shfl4f(var, tr.slot[t] * LPR + w.l); // the slot's row slice
        float4 c;
        if (MODE == 0) {
            float x;
            const float s = triple_s<LPR>(dot4(own, tr.a[t]) - dot4(own, tr.b[t]), &x);
            const int pos = tr.rec[t].w & 0x7fffffff;
            if (tr.has[t] && w.l == 0)
                w.xb[pos] = x; // softplus(-x) is taken in bpr_reduce_loss
            if (STORE_CONTRIB && tr.has[t] && w.act)
                st4s(w.contrib + (int64_t)pos * w.d + 4 * w.l, make_float4(s * own.x, s * own.y, s * own.z, s * own.w));
            const float ns = -s;
            c.x = ns * (tr.a[t].x - tr.b[t].x) + w.reg * own.x;
            c.y = ns * (tr.a[t].y - tr.b[t].y) + w.reg * own.y;
            c.z = ns * (tr.a[t].z - tr.b[t].z) + w.reg * own.z;
            c.w = ns * (tr.a[t].w - tr.b[t].w) + w.reg * own.w;
        } else {
            const bool is_j = tr.rec[t].w < 0;
            const float sg = is_j ? 1.0f : -1.0f; // j-slices get +s*p, i-slices -s*p
            float4 sp;
            if (MODE == 1) {
                // p.q_i - p.q_j with the owned row on either side: (p.other - p.own) * sg, the sign flip is exact
                float x;
                const float s = triple_s<LPR>((dot4(tr.a[t], tr.b[t]) - dot4(tr.a[t], own)) * sg, &x);
                const float ss = sg * s;
                sp = make_float4(ss * tr.a[t].x, ss * tr.a[t].y, ss * tr.a[t].z, ss * tr.a[t].w);
            } else {
                sp = make_float4(sg * tr.a[t].x, sg * tr.a[t].y, sg * tr.a[t].z, sg * tr.a[t].w);
            }
            c.x = sp.x + w.reg * own.x;
            c.y = sp.y + w.reg * own.y;
            c.z = sp.z + w.reg * own.z;
            c.w = sp.w + w.reg * own.w;
        }
        // branch-free: a contribution that is not taken is added as +0 (exact)
        const bool take = tr.has[t] && w.act;
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const float k = (take && tr.slot[t] == s) ? 1.0f : 0.0f; // k * c is exact, so the fused form rounds once, the same
            acc[s] = make_float4(__builtin_fmaf(k, c.x, acc[s].x), __builtin_fmaf(k, c.y, acc[s].y),
                                 __builtin_fmaf(k, c.z, acc[s].z), __builtin_fmaf(k, c.w, acc[s].w));
        }
    }
}

// walk the occurrence records [k_begin, k_end) of this wave
template <int LPR, int RW, int MODE, bool STORE_CONTRIB>
__device__ __forceinline__ void walk_range(const WaveCtx<Geo<LPR, RW>::R> &w, float4 var, AccArr<LPR, RW> &acc,
                                           const int4 *__restrict__ occ, int k_begin, int k_end)
{
    constexpr int G = Geo<LPR, RW>::G;
    const int lane = threadIdx.x & 63;
    if ((CREC_ABL & 16) && k_end > k_begin + 16) // ablation (timing only): no wave walks more than 16 records
        k_end = k_begin + 16;
    for (int c0 = k_begin; c0 < k_end; c0 += 64) {
        const int nc = (k_end - c0) < 64 ? (k_end - c0) : 64;
        const int4 mine = lane < nc ? occ[c0 + lane] : make_int4(0, 0, 0, 0);
#ifndef CREC_STAMP_VAR
        if (c0 == k_begin)
            crec_mark(1, mine.x); // the first chunk of records has arrived
#endif
        const int n_sub = (nc + G - 1) / G;
        if (PIPELINED) {
            Trip ta, tb;
            trip_issue<LPR, RW, MODE>(ta, w, mine, c0, nc, n_sub, 0);
            for (int t0 = 0; t0 < n_sub; t0 += 2 * UNR) {
                if (t0 + UNR < n_sub)
                    trip_issue<LPR, RW, MODE>(tb, w, mine, c0, nc, n_sub, t0 + UNR);
                trip_consume<LPR, RW, MODE, STORE_CONTRIB>(ta, w, var, acc);
                if (t0 + UNR >= n_sub)
                    break;
                if (t0 + 2 * UNR < n_sub)
                    trip_issue<LPR, RW, MODE>(ta, w, mine, c0, nc, n_sub, t0 + 2 * UNR);
                trip_consume<LPR, RW, MODE, STORE_CONTRIB>(tb, w, var, acc);
            }
        } else {
            for (int t0 = 0; t0 < n_sub; t0 += UNR) {
                Trip ta;
                trip_issue<LPR, RW, MODE>(ta, w, mine, c0, nc, n_sub, t0);
                trip_consume<LPR, RW, MODE, STORE_CONTRIB>(ta, w, var, acc);
            }
        }
    }
}

template <int LPR, int RW, int MODE, bool STORE_CONTRIB>
__device__ __forceinline__ float4 slots_grad(float4 var, const Slot &sl, int n_valid, const float *__restrict__ P,
                                    const float *__restrict__ Q, int d, const int4 *__restrict__ occ,
                                    float *__restrict__ contrib, float *__restrict__ xb, const Hyper &h, int part = 0,
                                    int nparts = 1)
{
    constexpr int R = Geo<LPR, RW>::R;
    const int lane = threadIdx.x & 63;
    WaveCtx<R> w;
    w.P = P;
    w.Q = Q;
    w.contrib = contrib;
    w.xb = xb;
    w.d = d;
    w.g = lane / LPR;
    w.l = lane % LPR;
    w.act = Geo<LPR, RW>::FULL ? true : 4 * w.l < d;
    w.reg = h.reg;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    int st[R], cn[R]; // wave-uniform copies of the slots' segments
#pragma unroll
    for (int s = 0; s < R; ++s) {
        st[s] = __builtin_amdgcn_readlane(sl.start, s * LPR); // a fixed lane: a scalar, not a cross-lane permute
        cn[s] = s < n_valid ? __builtin_amdgcn_readlane(sl.cnt, s * LPR) : 0;
        const int len = cn[s] < 0 ? -cn[s] : cn[s];
        w.e[s] = s < n_valid ? st[s] + len : 0x7fffffff; // INT_MAX for slots beyond the table
    }
    float4 acc[R];
#pragma unroll
    for (int s = 0; s < R; ++s)
        acc[s] = zero;
    bool any_heavy = false;
    int first = 0x7fffffff, last = 0;
#pragma unroll
    for (int s = 0; s < R; ++s) {
        any_heavy |= cn[s] < 0;
        if (s < n_valid) {
            first = st[s] < first ? st[s] : first;
            last = w.e[s];
        }
    }
#ifdef CREC_STAMPS
    {
        const int wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if (lane == 0 && wv < 16384 && R > 1)
            crec_stamps[8 * wv + 7] = ((unsigned long long)(unsigned)(last - first) << 32) | (unsigned)(any_heavy ? 1 : 0) | 2u;
    }
#endif
    if (nparts > 1) { // a wide group: this wave walks its share of the range (whole sub-iterations of G records)
        constexpr int G = Geo<LPR, RW>::G;
        const int per = ((last - first + nparts - 1) / nparts + G - 1) / G * G;
        first += part * per;
        last = (first + per) < last ? (first + per) : last;
    }
    if (!any_heavy) { // the usual case: the slots' segments are one contiguous range
        if (last > first)
            walk_range<LPR, RW, MODE, STORE_CONTRIB>(w, var, acc, occ, first, last);
    } else { // a heavy slot sits in the middle: walk the other slots one by one
#pragma unroll 1
        for (int s = 0; s < n_valid; ++s) {
            int b = st[0], c = cn[0];
#pragma unroll
            for (int q = 1; q < R; ++q) {
                b = s == q ? st[q] : b;
                c = s == q ? cn[q] : c;
            }
            if (c > 0)
                walk_range<LPR, RW, MODE, STORE_CONTRIB>(w, var, acc, occ, b, b + c);
        }
    }
    return reduce_to_owner<LPR, RW>(acc);
}

// ------------------------------------------------------------------------------------
// Loss bookkeeping.  loss_b = sum_triples softplus(-x_b) + reg * sum_rows cnt_row * |row|^2 / 2:
// the l2 part of BPR.py:42-43 regrouped by table row (each row's norm is taken once by the group
// that owns the row, times the number of batch slots that index it).  User waves store x_b per
// triple, every owner group stores its row term; bpr_reduce_loss does the softplus and the sums.
// Per batch the workspace holds xb[B] then rowterm[RT] (RT = U + I in tf1 mode, 3B in lazy mode).
// ------------------------------------------------------------------------------------
template <int LPR, int MODE, bool WT>
__device__ __forceinline__ void row_finish(const Tables &dst, const Slot &sl, int64_t ro, bool owner, float4 var, float4 m,
                                           float4 v, float4 grad, float *__restrict__ rowterm_slot, bool dense,
                                           const Hyper &h);

template <int LPR, int MODE, bool STORE_CONTRIB, bool FULL, bool WT = false>
__device__ __forceinline__ void row_wave(const Tables &src, const Tables &dst, int d, const Slot &sl, int n_valid,
                                const int4 *__restrict__ occ, float *__restrict__ contrib, float *__restrict__ xb,
                                float *__restrict__ rowterm_slot, bool dense, const Hyper &h)
{
    constexpr int RW = ROWS_PER_WAVE | (FULL ? 0x100 : 0), R = Geo<LPR, RW>::R;
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const bool in_slot = g < R && g < n_valid && (FULL || 4 * l < d);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool user = MODE == 0;
    const int64_t ro = sl.row * d + 4 * l;
    // the rows are fetched without waiting for their segment {start, count} (sl, still in flight when this is called
    // with a dense table: the row ids follow from the wave id): one round trip less on every wave's critical path.
    // The few rows that turn out to be heavy or in a wide group are fetched for nothing.
    float4 var = zero, m = zero, v = zero;
    if (in_slot) {
        var = ld4((user ? src.P : src.Q) + ro);
        if (user || !h.dQ) { // the data-parallel step hands item rows' Adam to bpr_adam_dense: no need for their slots here
            m = ld4s((user ? src.mP : src.mQ) + ro);
            v = ld4s((user ? src.vP : src.vQ) + ro);
        }
    }
    const bool heavy = sl.cnt < 0;
    const bool owner = in_slot && !heavy; // heavy rows are finished by their last chunk wave
    crec_mark(0, sl.cnt); // the row's segment {start, count} has arrived
#ifdef CREC_STAMP_VAR
    crec_mark(1, __float_as_int(var.x)); // diagnosis: when the wave's own row arrives (overrides the records mark)
#endif
    const float4 grad = slots_grad<LPR, RW, MODE, STORE_CONTRIB>(var, sl, n_valid, src.P, src.Q, d, occ, contrib, xb, h);
    crec_mark(2, __float_as_int(grad.x)); // every record walked, the gradient reduced to its owner
    row_finish<LPR, MODE, WT>(dst, sl, ro, owner, var, m, v, grad, rowterm_slot, dense, h);
}

// the owner group's end of a row: row term, Adam (or the gradient row for the data-parallel all-reduce), stores
template <int LPR, int MODE, bool WT>
__device__ __forceinline__ void row_finish(const Tables &dst, const Slot &sl, int64_t ro, bool owner, float4 var, float4 m,
                                           float4 v, float4 grad, float *__restrict__ rowterm_slot, bool dense,
                                           const Hyper &h)
{
    constexpr bool user = MODE == 0;
    const int l = (threadIdx.x & 63) % LPR;
    const float nr = group_sum<LPR>(dot4(var, var));
    if (owner) {
        if (l == 0 && (dense || sl.cnt > 0))
            *rowterm_slot = (0.5f * nr) * (float)sl.cnt;
        if (!user && h.dQ) { // the all-reduce + bpr_adam_dense finish this row
            st4_dq(h.dQ, ro, grad, h.dq_sys, false);
        } else {
            adam4(var, m, v, grad, sl.cnt > 0, h);
            if (WT) {
                st4_wt(user ? dst.P : dst.Q, ro, var);
                st4_wt(user ? dst.mP : dst.mQ, ro, m);
                st4_wt(user ? dst.vP : dst.vQ, ro, v);
            } else {
                st4s((user ? dst.P : dst.Q) + ro, var);
                st4s((user ? dst.mP : dst.mQ) + ro, m);
                st4s((user ? dst.vP : dst.vQ) + ro, v);
            }
        }
    }
}

// A wide group (tf1 mode): the R rows of one row wave hold more occurrences than a single wave should walk (a launch
// lasts as long as its longest wave: 50 records take a row wave twice as long as the median one).  A whole workgroup
// takes the group: its NW waves walk equal shares of the group's record range with the usual slot bookkeeping, the
// per-slot partial gradients meet in LDS, and wave 0 adds them in wave order and finishes the rows.  A barrier and
// 4 KB of LDS instead of the chunk waves' partials in memory + ticket, which cost as much as the walk they save.
template <int LPR, int MODE, bool FULL, bool WT>
__device__ __forceinline__ void wide_group(const Tables &src, const Tables &dst, int d, int64_t n_rows, int64_t group,
                                           const int2 *__restrict__ rowseg, const int4 *__restrict__ occ,
                                           float *__restrict__ xb, float *__restrict__ rowterm, const Hyper &h,
                                           float4 *__restrict__ lds /* [NW][64] */)
{
    constexpr int RW = ROWS_PER_WAVE | (FULL ? 0x100 : 0), R = Geo<LPR, RW>::R;
    constexpr bool user = MODE == 0;
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = (int)(blockDim.x >> 6);
    const int64_t base = group * R;
    const int n_valid = (int)((n_rows - base) < R ? (n_rows - base) : R);
    Slot sl;
    sl.row = base + (g < n_valid ? g : 0);
    const int2 rs = rowseg[sl.row];
    sl.start = rs.x;
    sl.cnt = g < n_valid ? (rs.y < 0 ? -rs.y : rs.y) : 0; // the index flags the group's rows like heavy rows: undo
    const bool owner = g < R && g < n_valid && (FULL || 4 * l < d);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t ro = sl.row * d + 4 * l;
    float4 var = zero, m = zero, v = zero;
    if (owner) {
        var = ld4((user ? src.P : src.Q) + ro); // every wave: the records' arithmetic needs the rows
        if (wv == 0 && (user || !h.dQ)) {
            m = ld4s((user ? src.mP : src.mQ) + ro);
            v = ld4s((user ? src.vP : src.vQ) + ro);
        }
    }
    float4 grad = slots_grad<LPR, RW, MODE, false>(var, sl, n_valid, src.P, src.Q, d, occ, nullptr, xb, h, wv, nw);
    lds[wv * 64 + lane] = grad;
    __syncthreads();
    if (wv != 0)
        return;
    grad = lds[lane];
    for (int k = 1; k < nw; ++k) // wave order: the sum does not depend on timing
        add4(grad, lds[k * 64 + lane]);
    row_finish<LPR, MODE, WT>(dst, sl, ro, owner, var, m, v, grad, rowterm + sl.row, true, h);
}

// what the wave that finishes a heavy row needs beyond its own chunk
struct FinishArgs {
    const int4 *hrows; // this side's heavy row descriptors
    int32_t *tickets;  // this side's per-heavy-row counters, zero between launches
    Tables dst;        // tf1: the other buffer set; lazy: the tables themselves
    float *rowterm;    // this side's row terms
    int dense;         // row terms indexed by row id (tf1) or by rank among the batch's unique rows (lazy)
};

// one chunk (<= CH occurrences) of a heavy row: all G groups on the row, partial gradient out.  The chunk waves
// of a row take tickets; the wave that draws the last one adds the partials in chunk order (so the sum does not
// depend on which wave that is), applies Adam and writes the row term -- no second launch.  The partials cross
// workgroups, possibly XCDs: release fence before the ticket, acquire fence after it.
template <int LPR, int MODE, bool STORE_CONTRIB, bool FULL, bool WT = false>
__device__ __forceinline__ void chunk_wave(const Tables &src, int d, int4 item, const int4 *__restrict__ occ,
                                  float *__restrict__ contrib, float *__restrict__ xb, float *__restrict__ partial,
                                  const FinishArgs &f, const Hyper &h)
{
    constexpr bool user = MODE == 0 || MODE == 3;
    const int lane = threadIdx.x & 63, g = lane / LPR, l = lane % LPR;
    const bool act = FULL || 4 * l < d;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int hidx = item.z >> 8;
    Slot sl;
    sl.row = item.x;
    sl.start = item.y;
    sl.cnt = item.z & 0xff;
    const int64_t ro = sl.row * d + 4 * l;
    const bool on = g == 0 && act;
    float4 var = act ? ld4((user ? src.P : src.Q) + ro) : zero; // every group: R == 1
    // the finishing wave's inputs are fetched by every chunk wave up front: their latency hides behind the
    // chunk's own gathers instead of extending the row's critical path after the last ticket
    const int4 rec = f.hrows[hidx];
    const bool slots = on && (user || !h.dQ); // item rows that only emit their gradient row (h.dQ) have no slots here
    float4 m = slots ? ld4((user ? src.mP : src.mQ) + ro) : zero, v = slots ? ld4((user ? src.vP : src.vQ) + ro) : zero;
    float4 grad;
    if (MODE >= 2) {
        // lazy mode (item rows, and user rows whose contributions were stored by bpr_triples_lazy): an occurrence is one contribution row to read and add, nothing to compute first, so the
        // chunk keeps eight rows in flight per group instead of the generic walk's UNR (a chunk wave is the launch's
        // critical path).  Same per-group order and the same cross-group sum as the generic walk.
        constexpr int G = 64 / LPR, DEEP = 8;
        AccArr<LPR, 1 | (FULL ? 0x100 : 0)> acc;
        acc[0] = zero;
        for (int c0 = 0; c0 < sl.cnt; c0 += 64) {
            const int nc = (sl.cnt - c0) < 64 ? (sl.cnt - c0) : 64;
            const int mine = lane < nc ? occ[sl.start + c0 + lane].w : 0;
            const int n_sub = (nc + G - 1) / G;
            for (int t0 = 0; t0 < n_sub; t0 += DEEP) {
                float4 a[DEEP];
                int rw[DEEP];
                bool has[DEEP];
#pragma unroll
                for (int t = 0; t < DEEP; ++t) {
                    const int kk = (t0 + t) * G + g;
                    rw[t] = __shfl(mine, kk & 63, 64);
                    has[t] = (t0 + t) < n_sub && kk < nc;
                    a[t] = (has[t] && act) ? ld4(contrib + (int64_t)(rw[t] & 0x7fffffff) * d + 4 * l) : zero;
                }
#pragma unroll
                for (int t = 0; t < DEEP; ++t) {
                    const float sg = rw[t] < 0 ? 1.0f : -1.0f, k = (has[t] && act) ? 1.0f : 0.0f;
                    add4(acc[0], make_float4(k * (sg * a[t].x + h.reg * var.x), k * (sg * a[t].y + h.reg * var.y),
                                             k * (sg * a[t].z + h.reg * var.z), k * (sg * a[t].w + h.reg * var.w)));
                }
            }
        }
        grad = reduce_to_owner<LPR, 1 | (FULL ? 0x100 : 0)>(acc);
    } else {
        grad = slots_grad<LPR, 1 | (FULL ? 0x100 : 0), MODE, STORE_CONTRIB>(var, sl, 1, src.P, src.Q, d, occ, contrib, xb, h);
    }
    // publish the partial, then take a ticket.  The partial travels with agent-scope stores (served where the XCDs
    // meet) and the wave waits for their completion before its ticket: the same guarantee as a release fence for
    // this one buffer, without writing back the whole L2 (a fence pair per chunk wave cost ~4 us of the launch).
    if (CREC_TICKET_FENCES) {
        if (on)
            st4(partial + (int64_t)item.w * d + 4 * l, grad);
        __threadfence();
    } else {
        if (on)
            st4_agent(partial + (int64_t)item.w * d + 4 * l, grad);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    int ticket = 0;
    if (lane == 0)
        ticket = __hip_atomic_fetch_add(f.tickets + hidx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __builtin_amdgcn_readfirstlane(ticket);
    const int first = rec.y, parts = rec.z >> 16, cnt = rec.z & 0xffff;
    if (ticket != parts - 1)
        return;
    if (CREC_TICKET_FENCES)
        __threadfence();
    else
        asm volatile("" ::: "memory");
    // the partials, chunk c by group c % G, eight in flight per group; then the groups' sums in the fixed order of the
    // cross-group reduction -- one order whichever wave ends up here (a single group reading all partials in sequence
    // was the longest link of a heavy row's chain: 54 partials = 7 dependent rounds)
    {
        constexpr int G = 64 / LPR;
        AccArr<LPR, 1 | (FULL ? 0x100 : 0)> accp;
        accp[0] = zero;
        if (act) {
            const float *pp = partial + (int64_t)first * d + 4 * l;
            const auto ldp = [](const float *q) { return CREC_TICKET_FENCES ? ld4(q) : ld4_agent(q); };
            for (int p0 = g; p0 < parts; p0 += 8 * G) {
                float4 t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    t[k] = (p0 + k * G) < parts ? ldp(pp + (int64_t)(p0 + k * G) * d) : zero;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    add4(accp[0], t[k]);
            }
        }
        grad = reduce_to_owner<LPR, 1 | (FULL ? 0x100 : 0)>(accp);
    }
    const float nr = group_sum<LPR>(dot4(var, var));
    if (on) {
        if (l == 0 && f.rowterm) // lazy mode keeps its l2 terms per triple
            f.rowterm[f.dense ? sl.row : (int64_t)rec.w] = (0.5f * nr) * (float)cnt;
        if (!user && h.dQ) {
            st4_dq(h.dQ, ro, grad, h.dq_sys, false);
        } else {
            adam4(var, m, v, grad, true, h);
            if (WT) {
                st4_wt(user ? f.dst.P : f.dst.Q, ro, var);
                st4_wt(user ? f.dst.mP : f.dst.mQ, ro, m);
                st4_wt(user ? f.dst.vP : f.dst.vQ, ro, v);
            } else {
                st4((user ? f.dst.P : f.dst.Q) + ro, var);
                st4((user ? f.dst.mP : f.dst.mQ) + ro, m);
                st4((user ? f.dst.vP : f.dst.vQ) + ro, v);
            }
        }
    }
    if (lane == 0)
        __hip_atomic_store(f.tickets + hidx, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch
}

// ------------------------------------------------------------------------------------
// tf1 mode: the whole step in one launch, tables read from `src`, written to `dst`.
// Wave ids: [0, HIu) user chunk waves, [HIu, HIu+HIq) item chunk waves (first, so that the longest
// work starts first), then WU waves owning user rows, then WI waves owning item rows.
// ------------------------------------------------------------------------------------
struct HeavyArgs {
    const int32_t *hn;               // {user heavy rows, user chunks, item heavy rows, item chunks}
    const int4 *u_hitems, *q_hitems; // chunk descriptors
    const int4 *u_hrows, *q_hrows;   // heavy row descriptors
    float *partial;                  // [HIu + HIq][d]
    int32_t *tickets;                // [HRu max + HRq max], zero between launches
    int HIu, HIq, HRu, HRq;
    int HRu_max;                     // item tickets start here
    const int32_t *u_wide, *q_wide;  // wide groups of this batch (tf1 mode)
    int NWu, NWq, NWpad;
};

// the work of one wave id of a tf1 step: [0, HIu) user chunk waves, [HIu, HIu+HIq) item chunk waves, then WU waves
// owning user rows, then WI waves owning item rows
template <int LPR, bool FULL, bool WT = false>
__device__ __forceinline__ void fused_wave(int64_t wave_id, const Tables &src, const Tables &dst, int d, int64_t U, int64_t I,
                                           const int2 *__restrict__ u_rowseg, const int4 *__restrict__ u_occ,
                                           const int2 *__restrict__ q_rowseg, const int4 *__restrict__ q_occ,
                                           const HeavyArgs &hv, float *__restrict__ xb, float *__restrict__ rowterm,
                                           const Hyper &h)
{
    constexpr int R = Geo<LPR, ROWS_PER_WAVE>::R;
    const int g = (threadIdx.x & 63) / LPR;
    if (wave_id < hv.HIu) {
        chunk_wave<LPR, 0, false, FULL, WT>(src, d, hv.u_hitems[wave_id], u_occ, nullptr, xb, hv.partial,
                                        FinishArgs{hv.u_hrows, hv.tickets, dst, rowterm, 1}, h);
        return;
    }
    wave_id -= hv.HIu;
    if (wave_id < hv.HIq) {
        chunk_wave<LPR, 1, false, FULL, WT>(src, d, hv.q_hitems[wave_id], q_occ, nullptr, xb, hv.partial + (int64_t)hv.HIu * d,
                                        FinishArgs{hv.q_hrows, hv.tickets + hv.HRu_max, dst, rowterm + U, 1}, h);
        return;
    }
    wave_id -= hv.HIq;
    const int64_t WU = (U + R - 1) / R, WI = (I + R - 1) / R;
    if (wave_id >= WU + WI)
        return;
    const bool is_user = CREC_ITEMS_FIRST ? wave_id >= WI : wave_id < WU;
    const int64_t n_rows = is_user ? U : I;
    const int64_t base = (CREC_ITEMS_FIRST ? (is_user ? wave_id - WI : wave_id) : (is_user ? wave_id : wave_id - WU)) * R;
    const int n_valid = (int)((n_rows - base) < R ? (n_rows - base) : R);
    Slot sl;
    sl.row = base + (g < n_valid ? g : 0);
    const int2 rs = (is_user ? u_rowseg : q_rowseg)[sl.row];
    sl.start = rs.x;
    sl.cnt = g < n_valid ? rs.y : 0;
    float *rt = rowterm + (is_user ? 0 : U) + sl.row;
    if (is_user)
        row_wave<LPR, 0, false, FULL, WT>(src, dst, d, sl, n_valid, u_occ, nullptr, xb, rt, true, h);
    else
        row_wave<LPR, 1, false, FULL, WT>(src, dst, d, sl, n_valid, q_occ, nullptr, xb, rt, true, h);
}

template <int LPR, bool FULL>
__global__ __launch_bounds__(CREC_FUSED_BLOCK) void bpr_step_fused(Tables src, Tables dst, int d, int64_t U, int64_t I,
                                                      const int2 *__restrict__ u_rowseg,
                                                      const int4 *__restrict__ u_occ,
                                                      const int2 *__restrict__ q_rowseg,
                                                      const int4 *__restrict__ q_occ, HeavyArgs hv,
                                                      float *__restrict__ xb, float *__restrict__ rowterm, Hyper h)
{
    constexpr int R = Geo<LPR, ROWS_PER_WAVE>::R;
    const int lane = threadIdx.x & 63, g = lane / LPR;
    // the wave index is the same in every lane: say so (readfirstlane), and everything derived from it -- which rows,
    // how many, which side, loop bounds -- stays in scalar registers and scalar branches
    if (CREC_ABL & 4) // ablation: the launch with no work (dispatch + drain floor of this grid)
        return;
    // the first workgroups are the wide groups' (users, then items); the wave ids of the rest start after them
    __shared__ float4 wide_lds[(CREC_FUSED_BLOCK / 64) * 64];
    const int n_wide = hv.NWpad; // the wide groups' workgroups, padded by the host so that the grid size changes rarely
#ifdef CREC_STAMPS
    const unsigned long long t_begin = wall_clock64(), c_begin = clock64();
    const int raw_wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const auto stamp_end = [&](unsigned flag) {
        if (lane == 0 && raw_wave < 16384) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            crec_stamps[8 * raw_wave + 0] = t_begin;
            crec_stamps[8 * raw_wave + 1] = wall_clock64();
            crec_stamps[8 * raw_wave + 2] = clock64() - c_begin;
            crec_stamps[8 * raw_wave + 3] = xcc & 15;
            crec_stamps[8 * raw_wave + 7] |= flag;
        }
    };
#endif
    if ((int)blockIdx.x < n_wide) {
        if ((int)blockIdx.x < hv.NWu)
            wide_group<LPR, 0, FULL, CREC_FUSED_WT != 0>(src, dst, d, U, hv.u_wide[blockIdx.x], u_rowseg, u_occ, xb, rowterm, h,
                                                         wide_lds);
        else if ((int)blockIdx.x < hv.NWu + hv.NWq)
            wide_group<LPR, 1, FULL, CREC_FUSED_WT != 0>(src, dst, d, I, hv.q_wide[blockIdx.x - hv.NWu], q_rowseg, q_occ, xb,
                                                         rowterm + U, h, wide_lds);
#ifdef CREC_STAMPS
        stamp_end(4);
#endif
        return;
    }
    int64_t wave_id = (int64_t)(blockIdx.x - n_wide) * (blockDim.x >> 6) +
                      __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (CREC_ABL & 8) { // ablation: a pure copy of the six arrays, one wave per R rows (streaming floor of this grid)
        const int64_t WUc = (U + R - 1) / R, w2 = wave_id - hv.HIu - hv.HIq;
        const bool usr = w2 < WUc;
        const int64_t row = (usr ? w2 : w2 - WUc) * R + g;
        if (w2 >= 0 && g < R && row < (usr ? U : I)) {
            const int64_t ro = row * d + 4 * (lane % LPR);
            st4((usr ? dst.P : dst.Q) + ro, ld4((usr ? src.P : src.Q) + ro));
            st4((usr ? dst.mP : dst.mQ) + ro, ld4((usr ? src.mP : src.mQ) + ro));
            st4((usr ? dst.vP : dst.vQ) + ro, ld4((usr ? src.vP : src.vQ) + ro));
        }
    } else {
        fused_wave<LPR, FULL, CREC_FUSED_WT != 0>(wave_id, src, dst, d, U, I, u_rowseg, u_occ, q_rowseg, q_occ, hv, xb, rowterm, h);
    }
#ifdef CREC_STAMPS
    stamp_end(0);
#endif
}

// ------------------------------------------------------------------------------------
// bpr_steps_flow's row dealing by weight.  A step of that kernel lasts as long as its busiest workgroup, and a workgroup's work
// is the number of records of its rows -- popular items.  flow_count_rows counts a call's occurrences per row; the host deals the
// rows heaviest first, each to the workgroup with the least work so far that still has a free slot (flow_deal_host; a few per cent
// more workgroups than the rows need, so that a row as busy as a whole average workgroup ends up nearly alone in its own).  The
// tables are made ONCE per shape and kept by the ctx; they decide WHERE a row is worked on, never what is computed (a row's
// contributions are added in batch order wherever it sits): results do not depend on them, bit for bit (tests/test_gpu_flow.py).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void flow_count_rows(const int32_t *__restrict__ u, const int32_t *__restrict__ i,
                                                       const int32_t *__restrict__ j, int64_t n, int32_t *__restrict__ cu,
                                                       int32_t *__restrict__ cq)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) {
        atomicAdd(cu + u[k], 1);
        atomicAdd(cq + i[k], 1);
        atomicAdd(cq + j[k], 1);
    }
}
constexpr int FLOW_DEAL_MAX = 1 << 16; // rows per side the dealing takes (the ctx keeps the tables; made once per shape, on the host)

// ------------------------------------------------------------------------------------
// tf1 mode, ALL steps of a chunk in one launch: bpr_steps_flow (round 4; opt-in, option "flow").
//
// bpr_step_fused is bound by what surrounds its waves -- a dispatch per step, every row read from and written to memory per
// step, three dependent round trips per row wave -- not by bytes.  Here a lane group OWNS a table row for up to
// FLOW_MAX_STEPS consecutive steps and keeps the row and its Adam slots in registers; what the other workgroups need of it, the
// row's values after step k, it publishes once per step into a version history (hist[k][row], written through to where the
// XCDs meet), and the gathers of step k + 1 read that version of the rows they need.  There is no device-wide barrier and no flag:
// an unpublished slot holds a sentinel in every word, a gather that reads the sentinel asks again, so the steps of different rows
// overlap as far as the batches' own dependences allow -- a dataflow execution of exactly the sequential arithmetic.
//   * rows are dealt to workgroups round-robin (row r: workgroup r mod NWG, slot r div NWG), so that neighbouring popular
//     rows do not share a workgroup; the index kernel sorts by that position (IndexWs::flow_*), which makes the records of a
//     workgroup's rows ONE contiguous range of the sorted list
//   * a step of a workgroup: (A) its records are dealt to ALL its lane groups, whichever slot they belong to -- the slots' rows
//     and the records sit in LDS -- each group gathers its records' other rows, forms x, s and the record's contribution row and
//     leaves it in LDS; (B) every owner adds its slot's contribution rows in record (= batch) order, applies Adam to its
//     registers, publishes the row and puts it back into LDS for the next step.  Two workgroup barriers per step; every sum has
//     one order, whatever the timing and however the steps are cut into launches
//   * what does not depend on other workgroups is asked for ahead: a step's records two steps, its segments four steps before;
//     every wait for such a load stands BEFORE the step's publishing store (vector memory operations retire in issue order: a
//     wait behind a written-through store costs the store's round trip)
//   * every wave of the launch must be resident (the host checks the occupancy before it takes this path); a wave waits only
//     for rows whose owners are resident and never wait for IT within the same step, and every wait gives up after
//     FLOW_SPIN_LIMIT polls or when another wave has given up (FlowArgs::err, reported by the host as an error)
//   * two histories alternate between launches: a launch fills one and puts the sentinel back into the slots the launch
//     before it filled in the other (each owner its own rows, one slot per step), so no memset sits between launches
// ------------------------------------------------------------------------------------
constexpr int FLOW_MAX_STEPS = 64;
constexpr uint32_t FLOW_SENT = 0xffffffffu; // an unpublished row version: this in every 4-byte word; a published word is never this
                                            // (flow_publishable)
constexpr unsigned FLOW_SPIN_CHECK = 1u << 8, FLOW_SPIN_LIMIT = 1u << 20; // polls between looks at the give-up flag; polls before a wave gives up
                                                                           // itself (backed off to ~50 us each: about a minute)
struct FlowArgs {
    const int2 *u_rowseg, *q_rowseg; // [C][Upos], [C][Ipos]: {segment start, length} by POSITION (workgroup * rows per workgroup + slot)
    const int4 *u_occ, *q_occ;       // [C][B], [C][2B]
    int64_t Upos, Ipos, B;
    int nwg_u, nwg_q;                // workgroups per side = the modulus of the row dealing
    const int32_t *row_u, *row_q;    // position -> row (-1: nobody) under the dealing by weight; null: row = slot * nwg + workgroup
    int C;                           // steps of this launch
    float *histP, *histQ;            // [C - 1][U * d], [C - 1][I * d]: the versions after steps 1 .. C - 1, all sentinel on entry
    float *nextP, *nextQ;            // the other history: slots [0, dirty_next) get the sentinel back
    int dirty_next;
    int64_t hP, hQ;                  // floats per version
    float *loss_ws;                  // per step: x_b [B], then the row terms [U + I] (by row id)
    int64_t loss_stride;
    int32_t *err;
    int abl; // timing ablations (option flow_abl; WRONG results): 1 gathers do not wait for their versions, 2 gathers read the tables
             // (plain loads) instead of the versions, 4 nothing is published or reset
    float lr_t[FLOW_MAX_STEPS];
};

// a lane's 16 bytes of a row version are not there yet.  The owner writes them with ONE 16-byte store; such stores have been observed to
// land whole on gfx950 (MI355X_MICROARCH.md, hand-offs), and the first and the last word are looked at here, so a store that landed as
// two 8-byte halves in either order is still caught (integer logic, no short-circuit branches)
__device__ __forceinline__ bool flow_unset(float4 v)
{
    return ((int)(__float_as_uint(v.x) == FLOW_SENT) | (int)(__float_as_uint(v.w) == FLOW_SENT)) != 0;
}
__device__ __forceinline__ float4 flow_publishable(float4 v) // a NaN that happens to carry the sentinel's bits becomes the canonical one
{
    const auto fix = [](float x) { return __float_as_uint(x) == FLOW_SENT ? __uint_as_float(0x7fc00000u) : x; };
    return make_float4(fix(v.x), fix(v.y), fix(v.z), fix(v.w));
}

#ifndef CREC_FLOW_WAVES
#define CREC_FLOW_WAVES 4 // waves per SIMD bpr_steps_flow is compiled for: all of its waves must be resident at once
#endif
#ifndef CREC_FLOW_DEEP
#define CREC_FLOW_DEEP 4 // records a lane group has in flight (two gathered rows each)
#endif
#ifndef CREC_FLOW_ITEM_DIV
#define CREC_FLOW_ITEM_DIV 2 // an item workgroup holds this many times fewer rows than a user workgroup (an item row is ~3x as busy)
#endif
template <int LPR, int RDIV> struct FlowGeo {
    static constexpr int G = 64 / LPR;                                  // lane groups per wave
    static constexpr int R0 = G < ROWS_PER_WAVE ? G : ROWS_PER_WAVE;
    static constexpr int R = R0 / RDIV > 0 ? R0 / RDIV : 1;             // slots (rows) per wave
    static constexpr int WPB = CREC_FUSED_BLOCK / 64, RPB = R * WPB;    // slots per workgroup
    static constexpr int NG = G * WPB;                                  // lane groups per workgroup
    static constexpr int NREC_ = 16384 / (LPR * 16);                    // records per round: 16 KB of contribution rows
    static constexpr int NREC = NREC_ > 256 ? 256 : NREC_;
    static constexpr int SEGRING = 5;                                   // steps of segments kept in LDS (asked for four steps ahead)
    // LDS, in float4 units: the slots' rows, the round's contribution rows, two buffers of records, the ring of segments
    static constexpr int OFF_VAR = 0, OFF_CON = OFF_VAR + RPB * LPR, OFF_REC = OFF_CON + NREC * LPR, OFF_SEG = OFF_REC + 2 * NREC;
    static constexpr int LDS_F4 = OFF_SEG + (SEGRING * (RPB + 1) * 2 + 3) / 4 + 1;
};

// One workgroup of bpr_steps_flow (see the header above).  MODE 0: user rows (records {u, i, j, pos}: gathers q_i, q_j), 1: item rows
// (gathers p_u and the triple's other item row; bit 31 of pos: the owned row is the triple's j)
template <int LPR, int MODE, bool FULL>
__device__ __forceinline__ void flow_wave(const Tables &src, const Tables &dst, int d, int64_t U, int64_t n_rows, int j, int nwg,
                                          const int32_t *__restrict__ row_tab, const int2 *__restrict__ rowseg, int64_t pos_stride,
                                          const int4 *__restrict__ occ, int64_t occ_stride, float *hist_own, float *next_own,
                                          int64_t h_own, const FlowArgs &f, const Hyper &h, float4 *__restrict__ lds)
{
    using Geo_ = FlowGeo<LPR, MODE == 1 ? CREC_FLOW_ITEM_DIV : 1>;
    constexpr int G = Geo_::G, R = Geo_::R, RPB = Geo_::RPB, NG = Geo_::NG, NREC = Geo_::NREC, SEGRING = Geo_::SEGRING, DEEP = CREC_FLOW_DEEP;
    constexpr bool user = MODE == 0;
    const int tid = threadIdx.x, lane = tid & 63, g = lane / LPR, l = lane % LPR;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gi = wv * G + g; // this lane group among the workgroup's
    float4 *l_var = lds + Geo_::OFF_VAR, *l_con = lds + Geo_::OFF_CON;
    int4 *l_rec = reinterpret_cast<int4 *>(lds + Geo_::OFF_REC);
    int *l_seg = reinterpret_cast<int *>(lds + Geo_::OFF_SEG); // [SEGRING][RPB + 1][2]: {start, count} per slot; entry RPB: {end of the range, 0}
    // ownership: slot sw = wv * R + g (g < R) holds row sw * nwg + j, or what the dealing by weight put at that position
    const int sw = wv * R + (g < R ? g : 0);
    const int64_t row = row_tab ? (int64_t)row_tab[(int64_t)j * RPB + sw] : (int64_t)sw * nwg + j;
    const bool act = FULL || 4 * l < d;
    const bool owner = g < R && row >= 0 && row < n_rows && act;
    const int64_t ro = row * d + 4 * l;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 var = zero, m = zero, v = zero;
    if (owner) {
        var = ld4((user ? src.P : src.Q) + ro);
        m = ld4s((user ? src.mP : src.mQ) + ro);
        v = ld4s((user ? src.vP : src.vQ) + ro);
        l_var[sw * LPR + l] = var;
    }
    const int q_reset = (f.dirty_next + f.C - 1) / f.C; // slots of the other history an owner resets per step
    const crec_f4v sent = {__uint_as_float(FLOW_SENT), __uint_as_float(FLOW_SENT), __uint_as_float(FLOW_SENT), __uint_as_float(FLOW_SENT)};
    // ---- what does not depend on anybody, asked for ahead.  Segments: thread t < RPB holds slot t's {start, count} of steps
    // k + 3 and k + 4 (two register stages), steps k .. k + 2 sit in the LDS ring; records: thread t holds record t of step k + 2's
    // first round (one register stage), steps k and k + 1 sit in the two LDS buffers
    const int2 *seg_at = rowseg + (int64_t)j * RPB + (tid < RPB ? tid : RPB - 1);
    const int C_ = f.C;
    const auto seg_load = [&](int step) { return step < C_ ? seg_at[(int64_t)step * pos_stride] : make_int2(0, 0); };
    const auto seg_stage = [&](int step, int2 sg) {
        if (tid < RPB) {
            int *e = l_seg + ((step % SEGRING) * (RPB + 1) + tid) * 2;
            e[0] = sg.x, e[1] = sg.y;
            if (tid == RPB - 1)
                e[2] = sg.x + sg.y, e[3] = 0;
        }
    };
    const auto seg_first = [&](int step) { return l_seg[(step % SEGRING) * (RPB + 1) * 2]; };
    const auto seg_end = [&](int step) { return l_seg[((step % SEGRING) * (RPB + 1) + RPB) * 2]; };
    const auto rec_load = [&](int step) { // record tid of the step's range (its first round)
        int4 r = make_int4(0, 0, 0, 0);
        if (step < C_) {
            const int F = seg_first(step), L = seg_end(step);
            if (tid < NREC && tid < L - F)
                r = occ[(int64_t)step * occ_stride + F + tid];
        }
        return r;
    };
    seg_stage(0, seg_load(0));
    seg_stage(1, seg_load(1));
    seg_stage(2, seg_load(2));
    int2 seg_a = seg_load(3), seg_b = seg_load(4); // steps k + 3, k + 4
    __syncthreads();
    if (tid < NREC) {
        l_rec[tid] = rec_load(0);
        l_rec[NREC + tid] = rec_load(1);
    }
    int4 rec_a = rec_load(2); // step k + 2
    __syncthreads();
#ifdef CREC_STAMPS // tools/flow_stamps.py: phase times of every wave in one step of the launch ([wave][8], 100 MHz clock)
    const int stamp_wave = blockIdx.x * Geo_::WPB + wv;
#define CREC_FLOW_MARK(slot, dep)                                                                                     \
    do {                                                                                                              \
        asm volatile("" ::"v"(dep));                                                                                  \
        if (k == CREC_FLOW_STAMP_STEP && lane == 0 && stamp_wave < 16000)                                             \
            crec_stamps[8 * stamp_wave + (slot)] = wall_clock64();                                                    \
    } while (0)
#else
#define CREC_FLOW_MARK(slot, dep)
#endif
    // (the launch's arguments sit in kernel-argument memory: what a step needs of them is taken once, and the per-step pointers are
    // stepped instead of multiplied -- a scalar load and its wait per use was half a microsecond of every step)
    const int C = f.C, abl = f.abl, dirty_next = f.dirty_next;
    const int64_t hP = f.hP, hQ = f.hQ, loss_stride = f.loss_stride, Bx = f.B;
    const float *histP_k = f.histP - hP, *histQ_k = f.histQ - hQ; // version k - 1's slot as step k sees it
    const int4 *occ_k = occ;
    float *xb = f.loss_ws, *hist_k = hist_own;
    int32_t *const err = f.err;
    for (int k = 0; k < C; ++k, occ_k += occ_stride, xb += loss_stride, histP_k += hP, histQ_k += hQ, hist_k += h_own) {
        CREC_FLOW_MARK(0, 0);
        const bool hist = k > 0 && !(abl & 2);
        const float *Pk = hist ? histP_k : src.P;
        const float *Qk = hist ? histQ_k : src.Q;
        Hyper hk = h;
        hk.lr_t = f.lr_t[k];
        const int F = seg_first(k), L = seg_end(k), n_all = L - F; // the workgroup's records of this step
        const int4 *rec_buf = l_rec + (k & 1) * NREC;
        float4 grad = zero;
        const int my_start = l_seg[((k % SEGRING) * (RPB + 1) + sw) * 2] - F, my_cnt = g < R ? l_seg[((k % SEGRING) * (RPB + 1) + sw) * 2 + 1] : 0;
        for (int r0 = 0; r0 == 0 || r0 < n_all; r0 += NREC) { // rounds of NREC records (one, unless rows are very heavy)
            const int n_round = (n_all - r0) < NREC ? (n_all - r0) : NREC;
            if (r0 > 0) { // a further round: its records come straight from memory
                __syncthreads();
                if (tid < n_round)
                    l_rec[(k & 1) * NREC + tid] = occ_k[F + r0 + tid];
                __syncthreads();
            }
            // ---- (A) the round's records, dealt to all lane groups: record q of the round goes to group q mod NG
            for (int q0 = gi; q0 - g < n_round; q0 += NG * DEEP) { // (a wave goes on while its FIRST group has records: wave-uniform)
                int4 rec[DEEP];
                float4 a[DEEP], b[DEEP];
                bool has[DEEP];
                // the trip's depth: how many of its DEEP records the wave's first group still has -- wave-uniform, so that a wave with
                // one record per group (a user workgroup's usual step) does the arithmetic of one record, not of DEEP
                const int left = n_round - (q0 - g);
                const int nt = __builtin_amdgcn_readfirstlane(left >= DEEP * NG ? DEEP : (left + NG - 1) / NG);
                // every load of the trip is issued before the first one is waited for: NO per-lane condition around a load (the
                // compiler would wait for each before the next), dead groups read row 0 and drop it, lanes beyond d read lane 0's slice
                int64_t oa[DEEP], ob[DEEP];
#pragma unroll
                for (int t = 0; t < DEEP; ++t) {
                    const int q = q0 + t * NG;
                    has[t] = q < n_round;
                    rec[t] = make_int4(0, 0, 0, 0);
                    oa[t] = ob[t] = 0;
                    a[t] = b[t] = zero;
                    if (t < nt) {
                        rec[t] = rec_buf[has[t] ? q : 0];
                        const int ra = has[t] ? (user ? rec[t].y : rec[t].x) : 0, rb = has[t] ? (user ? rec[t].z : (rec[t].w < 0 ? rec[t].y : rec[t].z)) : 0;
                        oa[t] = (int64_t)ra * d + (act ? 4 * l : 0);
                        ob[t] = (int64_t)rb * d + (act ? 4 * l : 0);
                    }
                }
                const float *ta = user ? Qk : Pk;
                if (hist) {
#pragma unroll
                    for (int t = 0; t < DEEP; ++t) {
                        if (t < nt) {
                            a[t] = ld4_sc1(ta, oa[t]);
                            b[t] = ld4_sc1(Qk, ob[t]);
                        }
                    }
                } else {
#pragma unroll
                    for (int t = 0; t < DEEP; ++t) {
                        if (t < nt) {
                            a[t] = ld4_buf(ta, oa[t]);
                            b[t] = ld4_buf(Qk, ob[t]);
                        }
                    }
                }
                if (hist && !(abl & 1)) { // a trip that read the sentinel somewhere asks again until its rows' owners have published
                    unsigned spins = 0;
                    for (;;) {
                        uint64_t bad[DEEP], any = 0; // per record of the trip: the lanes that read the sentinel (wave-uniform masks)
#pragma unroll
                        for (int t = 0; t < DEEP; ++t) {
                            bad[t] = t < nt ? __ballot(has[t] && act && (flow_unset(a[t]) || flow_unset(b[t]))) : 0ull;
                            any |= bad[t];
                        }
                        if (any == 0)
                            break;
                        // back off: a row that is a fraction of a microsecond away is asked for again at once; a wave that has waited
                        // long is waiting for workgroups that are not on the chip yet (another stream's kernels hold their place), and
                        // 3 600 waves polling flat out would slow those kernels -- and so its own wait -- a hundredfold
                        if (spins < 16)
                            __builtin_amdgcn_s_sleep(1);
                        else if (spins < 64)
                            __builtin_amdgcn_s_sleep(16);
                        else
                            for (int z = 0; z < (spins < 256 ? 2 : 16); ++z)
                                __builtin_amdgcn_s_sleep(127);
#pragma unroll
                        for (int t = 0; t < DEEP; ++t) { // the records that are not there yet, by every lane (a scalar branch, no per-lane
                            if (bad[t] != 0) {           // condition around a load; a published row reads the same again)
                                a[t] = ld4_sc1(ta, oa[t]);
                                b[t] = ld4_sc1(Qk, ob[t]);
                            }
                        }
                        if ((++spins & (FLOW_SPIN_CHECK - 1)) == 0 &&
                            (spins >= FLOW_SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) {
                            if (lane == 0)
                                __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            break;
                        }
                    }
                }
#pragma unroll
                for (int t = 0; t < DEEP; ++t) {
                    if (t >= nt) // (wave-uniform)
                        continue;
                    // dead groups and lanes beyond d take part in the arithmetic with zeros
                    const bool live = has[t] && act;
                    const float4 at = make_float4(live ? a[t].x : 0.f, live ? a[t].y : 0.f, live ? a[t].z : 0.f, live ? a[t].w : 0.f);
                    const float4 bt = make_float4(live ? b[t].x : 0.f, live ? b[t].y : 0.f, live ? b[t].z : 0.f, live ? b[t].w : 0.f);
                    // the record's own row: the index kernel left its slot in bits 27 .. 30 of the record, its values sit in LDS
                    const int osw = (rec[t].w >> 27) & (RPB - 1);
                    const float4 own = act ? l_var[osw * LPR + l] : zero;
                    float4 c;
                    if (user) {
                        float x;
                        const float s_ = triple_s<LPR>(dot4(own, at) - dot4(own, bt), &x);
                        if (has[t] && l == 0)
                            xb[rec[t].w & 0x07ffffff] = x; // softplus(-x) is taken in bpr_reduce_loss
                        const float ns = -s_;
                        c.x = ns * (at.x - bt.x) + hk.reg * own.x;
                        c.y = ns * (at.y - bt.y) + hk.reg * own.y;
                        c.z = ns * (at.z - bt.z) + hk.reg * own.z;
                        c.w = ns * (at.w - bt.w) + hk.reg * own.w;
                    } else {
                        const float sg = rec[t].w < 0 ? 1.0f : -1.0f; // j-slices get +s*p, i-slices -s*p
                        float x;
                        const float s_ = triple_s<LPR>((dot4(at, bt) - dot4(at, own)) * sg, &x);
                        const float ss = sg * s_;
                        c.x = ss * at.x + hk.reg * own.x;
                        c.y = ss * at.y + hk.reg * own.y;
                        c.z = ss * at.z + hk.reg * own.z;
                        c.w = ss * at.w + hk.reg * own.w;
                    }
                    if (live)
                        l_con[(q0 + t * NG) * LPR + l] = c;
                }
            }
            CREC_FLOW_MARK(1, r0);
            __syncthreads();
            // the waits for what was asked for a step ago stand HERE, before this step's publishing store: the records of step k + 2
            // take this step's buffer (nobody reads it any more) and the segments of step k + 3 their ring slot
            if (r0 + NREC >= n_all) {
                if (tid < NREC)
                    l_rec[(k & 1) * NREC + tid] = rec_a;
                seg_stage(k + 3, seg_a);
                seg_a = seg_b;
            }
            // ---- (B) every owner: its slot's contribution rows of this round, in record order
            if (owner) {
                const int lo = my_start > r0 ? my_start : r0, hi = (my_start + my_cnt) < (r0 + n_round) ? (my_start + my_cnt) : (r0 + n_round);
                // (in record order, eight rows asked of LDS at a time: a load per add was a chain of ~100 cycles per record, and a popular
                // item's forty records made their workgroup the step's last)
                for (int q = lo; q < hi; q += 8) {
                    float4 c8[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        c8[e] = l_con[((q + e < hi ? q + e : lo) - r0) * LPR + l];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (q + e < hi)
                            add4(grad, c8[e]);
                }
            }
        }
        CREC_FLOW_MARK(2, __float_as_int(grad.x));
#ifdef CREC_STAMPS
        if (k == CREC_FLOW_STAMP_STEP && lane == 0 && stamp_wave < 16000)
            crec_stamps[8 * stamp_wave + 5] = ((unsigned long long)(unsigned)n_all << 32) | (user ? 0x100u : 0u);
#endif
        const float nr = group_sum<LPR>(dot4(var, var));
        if (owner) {
            if (l == 0)
                xb[Bx + (user ? 0 : U) + row] = (0.5f * nr) * (float)my_cnt;
            adam4(var, m, v, grad, my_cnt > 0, hk);
            if (k + 1 < C) {
                if (!(abl & 4))
                    st4_wt(hist_k, ro, flow_publishable(var));
                l_var[sw * LPR + l] = var;
            } else {
                st4s((user ? dst.P : dst.Q) + ro, var);
                st4s((user ? dst.mP : dst.mQ) + ro, m);
                st4s((user ? dst.vP : dst.vQ) + ro, v);
            }
            for (int sr = k * q_reset; sr < (k + 1) * q_reset && sr < dirty_next && !(abl & 4); ++sr)
                __builtin_nontemporal_store(sent, reinterpret_cast<crec_f4v *>(next_own + (int64_t)sr * h_own + ro));
        }
        // behind the publishing store: what later steps need (their waits stand in front of the NEXT step's publishing store)
        seg_b = seg_load(k + 5);
        __syncthreads(); // (the ring slot of step k + 3 and the rows in LDS are complete; this step's records are no longer read)
        rec_a = rec_load(k + 3);
        CREC_FLOW_MARK(3, __float_as_int(var.x));
    }
}

template <int LPR, bool FULL>
__global__ __launch_bounds__(CREC_FUSED_BLOCK) __attribute__((amdgpu_waves_per_eu(CREC_FLOW_WAVES, CREC_FLOW_WAVES))) void bpr_steps_flow(Tables src, Tables dst, int d, int64_t U, int64_t I, FlowArgs f, Hyper h)
{
    __shared__ float4 flow_lds[FlowGeo<LPR, 1>::LDS_F4]; // (the user side's layout is the larger one)
    if ((int)blockIdx.x < f.nwg_u)
        flow_wave<LPR, 0, FULL>(src, dst, d, U, U, (int)blockIdx.x, f.nwg_u, f.row_u, f.u_rowseg, f.Upos, f.u_occ, f.B, f.histP, f.nextP, f.hP, f,
                                h, flow_lds);
    else
        flow_wave<LPR, 1, FULL>(src, dst, d, U, I, (int)blockIdx.x - f.nwg_u, f.nwg_q, f.row_q, f.q_rowseg, f.Ipos, f.q_occ, 2 * f.B,
                                f.histQ, f.nextQ, f.hQ, f, h, flow_lds);
}

// ------------------------------------------------------------------------------------
// lazy mode: touched rows only, in place, one launch
// ------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------
// W consecutive batches of the lazy-family step in ONE launch (option lazy_window).  A launch of 6 144 triples is too
// short for the memory system (a bare gather-update launch of that size takes as long as the whole step); what a longer launch
// needs in order to stay EXACT is that a triple of batch b reads a row only after the last earlier batch of the window that
// touches the row has finished it.  lazy_window_deps finds that batch for every row of every triple (per other batch of the
// window one probe of its presence filter, a binary search in its sorted unique rows where the filter says yes); in the launch,
// whoever finishes a row writes it through and then leaves a stamp at (batch, side, segment start), and a triple with such a
// predecessor waits for that stamp before it asks for the row.  Batches are dispatched in order (blockIdx.y), so a wave only ever
// waits for workgroups dispatched before its own, which never wait for later ones: no residency requirement, no dead-lock.
// ------------------------------------------------------------------------------------
constexpr int LAZY_WIN_MAX = 16;
struct LazyWin {
    int W;                 // batches of this launch
    int64_t B;             // triples per batch = the stride of the per-batch index arrays
    int64_t n_left;        // triples from this launch's first batch to the end of the stream (the last batch may be short)
    int64_t loss_stride;   // floats between two batches' loss terms
    int64_t z_stride;      // rows (and counters) between two batches' contribution buffers: 6 B
    const int4 *deps;      // [W][B] by processing slot (IndexWs::wdeps of the launch's first batch)
    int32_t *done;         // [LAZY_WIN_MAX][3 B] stamps by (batch in window, segment start; user side first, item side from B)
    int32_t stamp0;        // the stamp of the launch's first batch (a ctx-wide launch counter: never repeats)
    int32_t step0;         // deferred Adam: the optimizer step of the launch's first batch
    int32_t *err;          // give-up flag (page-locked)
    bool fit_p, fit_q;     // a user / item table lies within 4 GiB
    int abl;               // timing ablations (option flow_abl; WRONG results): 1 nobody waits for a predecessor
    float lr_t[LAZY_WIN_MAX];
};

__global__ __launch_bounds__(256) void lazy_window_deps(IndexWs w, int W)
{
    const int b = blockIdx.y, slot = blockIdx.x * 256 + threadIdx.x;
    const int64_t B = w.B;
    if (slot >= w.nslots[b * 2])
        return;
    const int w0 = b - b % W;
    const int2 pu = w.slot_pu[(int64_t)b * B + slot];
    const uint32_t rows[3] = {(uint32_t)pu.y, (uint32_t)w.i_desc[(int64_t)b * B + slot].x & 0x7fffffffu,
                              (uint32_t)w.j_desc[(int64_t)b * B + slot].x & 0x7fffffffu};
    int dep[3] = {-1, -1, -1};
    // the filters of all earlier batches of the window are probed at once (b and w0 are the workgroup's: the same trip count for every
    // lane), nearest batch first in the mask; only a set bit costs a binary search (one in eleven of them a false alarm)
    uint32_t hs[3], may[3] = {0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 3; ++r)
        hs[r] = wf_hash(rows[r]);
    for (int bp = b - 1, k = 0; bp >= w0; --bp, ++k) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const uint32_t word = w.wfilt[((int64_t)bp * 2 + (r ? 1 : 0)) * WF_WORDS + (hs[r] >> 5)];
            may[r] |= ((word >> (hs[r] & 31u)) & 1u) << k;
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int side = r ? 1 : 0;
        const uint32_t row = rows[r];
        while (may[r] != 0u && dep[r] < 0) { // the nearest earlier batch of the window that names the row
            const int k = __builtin_ctz(may[r]);
            may[r] &= may[r] - 1u;
            const int bp = b - 1 - k;
            const int32_t *wr = side ? w.wrow_q + (int64_t)bp * 2 * B : w.wrow_u + (int64_t)bp * B;
            int lo = 0, hi = w.wn[bp * 2 + side];
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if ((uint32_t)wr[mid] < row)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            if (lo < w.wn[bp * 2 + side] && (uint32_t)wr[lo] == row)
                dep[r] = ((bp - w0) << 20) | (side ? w.wst_q + (int64_t)bp * 2 * B : w.wst_u + (int64_t)bp * B)[lo];
        }
    }
    w.wdeps[(int64_t)b * B + slot] = make_int4(dep[0], dep[1], dep[2], 0);
}

struct LazyBufs {
    float *ucontrib, *qcontrib; // [B][d], [2B][d] contribution rows by sorted slot position (sign and s_b applied)
    float *upart, *qpart;       // [B][d], [2B][d] partial sums of heavy rows' chunks, at the chunk's first slot position
    int32_t *urow_t, *uchk_t;   // [B] ticket counters by segment start / chunk start, zero between launches
    int32_t *qrow_t, *qchk_t;   // [2B]
    int32_t CH;                 // slots per chunk of a heavy row
    float *wpart;               // [2B][d] polled partials of front rows without negative slots (lazy_poll_front): all bits set where nothing waits; null: tickets only
    int32_t *err;               // give-up flag of a wait inside the launch (page-locked)
};

// sum of n rows base[0], base[stride], ... in that order, each plus `add` (reg * row, or zero); agent-scope loads
// (the rows were published by other waves, possibly on another XCD), eight in flight
__device__ __forceinline__ float4 lazy_sum_rows(const float *buf, int64_t off, int n, int64_t stride, float4 add)
{
    constexpr int DEEP = CREC_LAZY_DEEP;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < n; k0 += DEEP) {
        float4 c[DEEP];
#pragma unroll
        for (int e = 0; e < DEEP; ++e)
            c[e] = k0 + e < n ? ld4_pub(buf, off + (int64_t)(k0 + e) * stride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int e = 0; e < DEEP; ++e)
            if (k0 + e < n)
                add4(g, make_float4(c[e].x + add.x, c[e].y + add.y, c[e].z + add.z, c[e].w + add.w));
    }
    return g;
}
// the same with the whole 4 GiB a 32-bit offset spans (lazy_window: a finisher's slots, where the table fits)
__device__ __forceinline__ float4 ld4_sc1w(const float *base, int64_t off)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, (int)0xffffffffu, 0x00020000);
    const crec_u4v x = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(uint32_t)((uint64_t)off * 4), 0, 16 /* sc1 */);
    return make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w));
}

// the sum of a front row: the partials of its pieces (l2 terms inside) -- the first piece ends at the workgroup boundary
// behind the row's first slot, the others are whole workgroups --, then its negative slots' contributions (+ reg*row each),
// all in ONE list of loads, in that order
template <int DEEP = CREC_LAZY_DEEP>
__device__ __forceinline__ float4 lazy_sum_front(const float *buf, int d, int l, int start, int ci, int cnt, int hmod, int SPW, float4 regrow)
{
    const int room = SPW - hmod, len1 = ci < room ? ci : room, np = 1 + (ci - len1 + SPW - 1) / SPW, n = np + (cnt - ci);
    const auto at = [&](int e) { return e == 0 ? start : (e < np ? start + len1 + (e - 1) * SPW : start + ci + (e - np)); };
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < n; k0 += DEEP) {
        float4 c[DEEP];
#pragma unroll
        for (int e = 0; e < DEEP; ++e)
            c[e] = k0 + e < n ? ld4_pub(buf, (int64_t)at(k0 + e) * d + 4 * l) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int e = 0; e < DEEP; ++e)
            if (k0 + e < n) {
                const float w = k0 + e >= np ? 1.0f : 0.0f; // w * regrow is exact
                add4(g, make_float4(c[e].x + w * regrow.x, c[e].y + w * regrow.y, c[e].z + w * regrow.z, c[e].w + w * regrow.w));
            }
    }
    return g;
}

// A slot's ticket: which counter it is drawn on, what the slot weighs and at which total the counter is complete.
//   front item row (ci > 0: two or more positive slots, consecutive processing slots in the item-sorted front region): the
//     row's counter; the leader of a piece (the row's slots inside one workgroup) weighs the piece, a negative slot weighs
//     one; complete at cnt
//   light row (cnt <= CH): the row's counter, one per slot, complete at cnt
//   other rows (user rows with more than CH slots, item rows whose slots are mostly negatives): the counter of the slot's
//     chunk of CH slots, complete at the chunk's length; the chunk's finisher then draws on the row's counter, one per chunk
struct LazyTicket {
    int idx, need;
    bool chunked;
};
__device__ __forceinline__ LazyTicket lazy_ticket_of(int4 desc, int CH)
{
    const int k = lazy_desc_k(desc), start = lazy_desc_start(desc), cnt = lazy_desc_cnt(desc), ci = lazy_desc_ci(desc);
    LazyTicket q;
    q.chunked = ci == 0 && cnt > CH;
    q.idx = q.chunked ? start + (k - start) / CH * CH : start;
    q.need = q.chunked ? ((cnt - (q.idx - start)) < CH ? (cnt - (q.idx - start)) : CH) : cnt;
    return q;
}

// the end of a multi-occurrence row: gradient sum g (l2 terms included) -> the optimizer's update in place; the row-sharded
// step sends an item row's sum back to its owner instead; deferred Adam first replays the steps the row sat out
template <int LPR, bool USER, bool DEFER>
__device__ __forceinline__ void lazy_finish_row(const Tables &t, int d, bool act, int row, float4 own, float4 g, float4 m, float4 v,
                                                const Hyper &h, int seg_start = 0)
{
    const int l = (threadIdx.x & 63) % LPR;
    int32_t *const win_done = USER ? h.win_done_u : h.win_done_q;
    if (win_done) { // lazy_window: the row written through, then -- once the stores have arrived -- this batch's stamp for the row's
        if (act) {  // next readers in the window (every lane of the group is here: the wait is the wave's)
            const int64_t ro = (int64_t)row * d + 4 * l;
            float4 nv = own;
            if (DEFER) { // m, v as stored: the steps the row sat out (var plays no part in their replay: any copy of the row will do)
                float4 scratch = own;
                int32_t *last = (USER ? h.lastP : h.lastQ) + row;
                adam_replay(scratch, m, v, __hip_atomic_load(last, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), h.step - 1, h);
                if (l == 0)
                    __hip_atomic_store(last, h.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            opt4(nv, m, v, g, h);
            st4_wtg((USER ? t.P : t.Q) + ro, nv);
            if (h.opt != CREC_OPT_SGD)
                st4_wtg((USER ? t.mP : t.mQ) + ro, m);
            if (h.opt == 0)
                st4_wtg((USER ? t.vP : t.vQ) + ro, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (l == 0) {
            __hip_atomic_store(win_done + seg_start, h.win_stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((seg_start & 15) == 0)
                win_t(h.win_wb, 8, false);
            if (!USER && seg_start == 0)
                win_t(h.win_wb, 7, false);
        }
        return;
    }
    if (!act)
        return;
    const int64_t ro = (int64_t)row * d + 4 * l;
    if (!USER && h.dQ) {
        st4_dq(h.dQ, ro, g, h.dq_sys, false);
        return;
    }
    float4 nv = own; // deferred mode: `own` is the contributor's caught-up copy of the row; m, v as stored are replayed here
    if (DEFER) {
        float4 scratch = ld4((USER ? t.P : t.Q) + ro);
        int32_t *last = (USER ? h.lastP : h.lastQ) + row;
        adam_replay(scratch, m, v, *last, h.step - 1, h);
        if (l == 0)
            *last = h.step;
    }
    opt4(nv, m, v, g, h);
    st4((USER ? t.P : t.Q) + ro, nv);
    if (h.opt != CREC_OPT_SGD)
        st4((USER ? t.mP : t.mQ) + ro, m);
    if (h.opt == 0)
        st4((USER ? t.vP : t.vQ) + ro, v);
}

// One multi-occurrence row of this group's triple.  The group has published its contribution (or, as the leader of a heavy
// block, the block's partial sum) at its slot's sorted position, waited for the store, and drawn a ticket: `old` = the
// counter before, `weight` = what it added.  Whoever completes the counter forms the row's (chunk's) sum in sorted order
// and finishes the row.  Every lane of the group takes the same path; nobody waits for anybody.
template <int LPR, bool USER, bool DEFER, bool WIN = false>
__device__ __forceinline__ void lazy_role_tail(const Tables &t, int d, bool act, int4 desc, int old, int weight, float4 own,
                                               float *cbuf, float *pbuf, int32_t *row_t, int32_t *chk_t, int CH, int SPW,
                                               const Hyper &h)
{
    const int lane = threadIdx.x & 63, l = lane % LPR, lead = lane - l;
    const int start = lazy_desc_start(desc), cnt = lazy_desc_cnt(desc), ci = lazy_desc_ci(desc);
    const LazyTicket q = lazy_ticket_of(desc, CH);
    if (!USER && start == 0 && l == 0 && (USER ? h.win_done_u : h.win_done_q) != nullptr) {
        win_t(h.win_wb, 9, true);
        win_t(h.win_wb, 10, false);
    }
    if (old + weight != q.need)
        return;
    pub_acquire();
    if (!USER && start == 0 && l == 0 && (USER ? h.win_done_u : h.win_done_q) != nullptr)
        win_t(h.win_wb, q.chunked ? 11 : 5, false);
    const int row = desc.x & 0x7fffffff;
    const int64_t ro = (int64_t)row * d + 4 * l;
    const bool slots = act && (USER || !h.dQ);
    // the slots are only needed at the very end: asked for now, their round trip hides behind the contributions'
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 m = zero, v = zero;
    if ((USER ? h.win_done_u : h.win_done_q) != nullptr) { // lazy_window: every load of a table row is an agent-scope one
        const bool wm = slots && h.opt != CREC_OPT_SGD, wv = slots && h.opt == 0;
        const float *safe = USER ? t.P : t.Q; // (a lane that wants nothing reads the table's first bytes)
        if (USER ? h.win_fit_p : h.win_fit_q) { // (a table within a descriptor's reach: loads the compiler tracks, in flight beside the sum below)
            if (wm)
                m = ld4_sc1w(USER ? t.mP : t.mQ, ro);
            if (wv)
                v = ld4_sc1w(USER ? t.vP : t.vQ, ro);
        } else {
            float4 rm, rv;
            win_load2(wm ? (USER ? t.mP : t.mQ) + ro : safe, wv ? (USER ? t.vP : t.vQ) + ro : safe, rm, rv);
            m = wm ? rm : zero, v = wv ? rv : zero;
        }
    } else {
        m = slots && h.opt != CREC_OPT_SGD ? ld4((USER ? t.mP : t.mQ) + ro) : zero;
        v = slots && h.opt == 0 ? ld4((USER ? t.vP : t.vQ) + ro) : zero;
    }
    const float4 regrow = make_float4(h.reg * own.x, h.reg * own.y, h.reg * own.z, h.reg * own.w);
    int32_t *ctr = (q.chunked ? chk_t : row_t) + q.idx;
    float4 g = zero;
    if (ci > 0) { // a front row: one partial per piece of its positive slots (l2 terms inside) -- the first piece ends at the
        // workgroup boundary behind the row's first slot, the others are whole workgroups --, then its negative slots
        if (act)
            g = lazy_sum_front<WIN ? CREC_WIN_DEEP : CREC_LAZY_DEEP>(cbuf, d, l, start, ci, cnt, lazy_desc_hmod(desc), SPW, regrow);
    } else if (act) {
        g = lazy_sum_rows(cbuf, (int64_t)q.idx * d + 4 * l, q.need, d, regrow);
    }
    if (l == 0)
        __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch
    if (q.chunked) { // a chunk of a chunked row: publish the partial, then the same game over the row's chunks
        if (act)
            st4_pub(pbuf, (int64_t)q.idx * d + 4 * l, g);
        pub_release();
        const int parts = (cnt + CH - 1) / CH;
        int t2 = -1;
        if (l == 0)
            t2 = __hip_atomic_fetch_add(row_t + start, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        t2 = __shfl(t2, lead, 64);
        if (t2 != parts - 1)
            return;
        pub_acquire();
        if (act)
            g = lazy_sum_rows(pbuf, (int64_t)start * d + 4 * l, parts, (int64_t)CH * d, zero);
        if (l == 0)
            __hip_atomic_store(row_t + start, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!USER && start == 0 && l == 0 && (USER ? h.win_done_u : h.win_done_q) != nullptr)
        win_t(h.win_wb, 6, false);
    lazy_finish_row<LPR, USER, DEFER>(t, d, act, row, own, g, m, v, h, start);
}

// lazy_window, a front row without negative slots (every slot of the row is one of its pieces' positive slots): no tickets.  The
// leaders of all pieces but the last leave their partial in `wpart` (at the piece's first sorted position; the buffer holds all bits
// set wherever no partial waits) and go; the LAST piece's leader -- in the workgroup dispatched last among the row's -- asks for the
// others' partials until none of them is the sentinel, adds them in piece order and its own at the end (the order lazy_sum_front adds
// them in), finishes the row and hands the entries back as sentinels.  Two trips through the fabric fewer on the row's way from
// batch to batch than publish -> acknowledgement -> ticket -> sum: the hottest row's chain is what bounds a window launch.
// It waits only for workgroups dispatched before its own, which wait for nobody dispatched later.
template <int LPR, bool DEFER>
__device__ __forceinline__ void lazy_poll_front(const Tables &t, int d, bool act, int4 desc, float4 own_row, float4 own_sum, float *wpart,
                                                int SPW, const Hyper &h, int32_t *err)
{
    constexpr int DEEP = 8; // (twelve spill in the 1024-thread geometry)
    const int lane = threadIdx.x & 63, l = lane % LPR;
    const int start = lazy_desc_start(desc), ci = lazy_desc_ci(desc), hmod = lazy_desc_hmod(desc);
    const int room = SPW - hmod, len1 = ci < room ? ci : room, np = 1 + (ci - len1 + SPW - 1) / SPW, n = np - 1; // the others' partials
    const auto at = [&](int e) { return e == 0 ? start : start + len1 + (e - 1) * SPW; };
    const int row = desc.x & 0x7fffffff;
    const int64_t ro = (int64_t)row * d + 4 * l;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool slots = act;
    float4 m = zero, v = zero;
    { // the row's slots: asked for now, in flight beside the wait for the partials
        const bool wm = slots && h.opt != CREC_OPT_SGD, wv = slots && h.opt == 0;
        if (h.win_done_q == nullptr) { // a launch per batch: nobody rewrites a row inside the launch
            if (wm)
                m = ld4(t.mQ + ro);
            if (wv)
                v = ld4(t.vQ + ro);
        } else if (h.win_fit_q) {
            if (wm)
                m = ld4_sc1w(t.mQ, ro);
            if (wv)
                v = ld4_sc1w(t.vQ, ro);
        } else {
            float4 rm, rv;
            win_load2(wm ? t.mQ + ro : t.Q, wv ? t.vQ + ro : t.Q, rm, rv);
            m = wm ? rm : zero, v = wv ? rv : zero;
        }
    }
    // the lanes of this group (the wave's other group may be elsewhere): the group's lanes agree on "all here" by ballot
    const uint64_t gmask = LPR == 64 ? ~0ull : (((1ull << LPR) - 1ull) << (lane - l));
    float4 g = zero;
    unsigned spins = 0;
    for (int k0 = 0; k0 < n; k0 += DEEP) {
        float4 c[DEEP];
        for (;;) {
            bool missing = false;
#pragma unroll
            for (int e = 0; e < DEEP; ++e)
                c[e] = (act && k0 + e < n) ? ld4_pub(wpart, (int64_t)at(k0 + e) * d + 4 * l) : zero;
#pragma unroll
            for (int e = 0; e < DEEP; ++e)
                missing |= act && k0 + e < n && __float_as_uint(c[e].x) == 0xffffffffu;
            if ((__builtin_amdgcn_ballot_w64(missing) & gmask) == 0)
                break;
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 255u) == 0 && (spins >= (1u << 20) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) {
                if (l == 0)
                    __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
#pragma unroll
        for (int e = 0; e < DEEP; ++e)
            if (k0 + e < n)
                add4(g, c[e]);
    }
    add4(g, own_sum);
    if (act) { // the entries go back to "nothing waits" (nobody asks for them again in this launch; the next launch starts after this one's end)
        const float4 sent = make_float4(__uint_as_float(0xffffffffu), __uint_as_float(0xffffffffu), __uint_as_float(0xffffffffu), __uint_as_float(0xffffffffu));
        for (int e = 0; e < n; ++e)
            st4_pub(wpart, (int64_t)at(e) * d + 4 * l, sent);
    }
    lazy_finish_row<LPR, false, DEFER>(t, d, act, row, own_row, g, m, v, h, start);
}

// bpr_triples_lazy: a lane group per processing slot; a workgroup = SPW slots (IndexWs::order, ::nslots); ud / id / jd: the
// triples' three roles by batch position.
template <int LPR, bool BIG = false> struct LazyGeo { // BIG: the geometry of batches >= 4096 triples (lazy_spw)
    static constexpr int SPW = LPR == 1 ? 64 : (LPR == 64 ? 16 : (BIG && LPR >= 8 ? 32 : (LPR == 32 && CREC_LAZY_SPW > 32 ? 32 : CREC_LAZY_SPW))),
                         THREADS = SPW * LPR; // = lazy_spw(d, B)
};

template <int LPR, bool FULL, bool DEFER, bool WIN = false, bool BIG = false>
__global__ __launch_bounds__((LazyGeo<LPR, BIG>::THREADS)) void bpr_triples_lazy(Tables t, int d, const int2 *__restrict__ slot_pu, int n_slots,
                                                                          const int32_t *__restrict__ nslots,
                                                                          const int4 *__restrict__ ud, const int4 *__restrict__ id,
                                                                          const int4 *__restrict__ jd, LazyBufs z,
                                                                          float *__restrict__ xb, float *__restrict__ l2b, Hyper h, const LazyWin win)
{
    constexpr int SPW = LazyGeo<LPR, BIG>::SPW;
    extern __shared__ float4 blk_rows[]; // [SPW][LPR], workgroups of the front region only
    const int lane = threadIdx.x & 63, l = lane % LPR, gw = (int)threadIdx.x / LPR;
    const int slot0 = (int)blockIdx.x * SPW;
    if (WIN) { // batch blockIdx.y of the launch's window: its slices of every per-batch array, its learning rate, its stamp
        const int wb = blockIdx.y;
        const int64_t left = win.n_left - (int64_t)wb * win.B;
        n_slots = (int)(left < win.B ? left : win.B);
        slot_pu += (int64_t)wb * win.B, ud += (int64_t)wb * win.B, id += (int64_t)wb * win.B, jd += (int64_t)wb * win.B;
        nslots += wb * 2;
        xb += (int64_t)wb * win.loss_stride, l2b += (int64_t)wb * win.loss_stride;
        const int64_t zr = (int64_t)wb * win.z_stride * d, zc = (int64_t)wb * win.z_stride;
        z.ucontrib += zr, z.qcontrib += zr, z.upart += zr, z.qpart += zr;
        if (z.wpart)
            z.wpart += (int64_t)wb * 2 * win.B * d;
        z.urow_t += zc, z.uchk_t += zc, z.qrow_t += zc, z.qchk_t += zc;
        h.lr_t = win.lr_t[wb];
        if (DEFER)
            h.step = win.step0 + wb;
        h.win_done_u = (win.abl & 2) ? nullptr : win.done + (int64_t)wb * 3 * win.B;
        h.win_done_q = (win.abl & 2) ? nullptr : h.win_done_u + win.B;
        h.win_stamp = win.stamp0 + wb;
        h.win_wb = wb;
        h.win_fit_p = win.fit_p, h.win_fit_q = win.fit_q;
        if (threadIdx.x == 0 && (blockIdx.x & 15) == 0)
            win_t(wb, 0, true);
    }
    const bool win_wt = WIN && !(win.abl & 2), win_sc1 = WIN && !(win.abl & 4);
    // ONE round of coalesced loads, by processing slot, gives every address of the next round: {position, user row} and the
    // two item descriptors.  (n_slots = the batch's triples comes from the host; the size of the front region and the
    // user's own descriptor -- needed only where the user occurs more than once -- arrive alongside.)
    const int n_blk_slots = nslots[1];
    const int slot = slot0 + gw;
    const bool valid = slot < n_slots, act = FULL || 4 * l < d, on = valid && act;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int4 none = make_int4((int)0x80000000u, 0, 0, 1); // a dead group: "single" rows, no tails
    const int2 pu = valid ? slot_pu[slot] : make_int2(0, 0);
    const int4 di = valid ? id[slot] : none, dj = valid ? jd[slot] : none;
    const int pos = pu.x;
    const bool front_block = slot0 < n_blk_slots; // workgroup-uniform: slots of the item-sorted front region in this workgroup
    const bool front = valid && slot < n_blk_slots; // this slot's positive item row has its positive slots right here, in a row
    const int4 du = (valid && !(CREC_LAZY_ABL & 16)) ? ud[pos] : none; // (a second, dependent load: off the critical path of the row loads below)
    const bool is = di.x < 0, js = dj.x < 0;
    const int64_t ou = (int64_t)pu.y * d + 4 * l, oi = (int64_t)(di.x & 0x7fffffff) * d + 4 * l,
                  oj = (int64_t)(dj.x & 0x7fffffff) * d + 4 * l;
    const bool item_adam = h.dQ == nullptr; // the row-sharded step only emits the item rows' gradient sums
    // front region: the piece of this slot's row inside this workgroup starts at the row's first slot or at the workgroup's;
    // the group of that slot leads the piece.  A piece that is the whole row is finished on the spot: its leader wants the slots
    const int i_k = lazy_desc_k(di), i_posr = i_k - lazy_desc_start(di), i_ci = lazy_desc_ci(di), i_cnt = lazy_desc_cnt(di);
    const bool blk_lead = front && (gw == 0 || i_posr == 0);
    const int blk_n = blk_lead ? ((i_ci - i_posr) < (SPW - gw) ? (i_ci - i_posr) : (SPW - gw)) : 0; // the piece's slots = the leader's weight
    const bool blk_whole = blk_lead && blk_n == i_cnt; // the row lies inside this workgroup: no ticket, no memory round trip
    float4 p = zero, qi = zero, qj = zero, mp = zero, vp = zero, mi = zero, vi = zero, mj = zero, vj = zero;
    int4 dp = make_int4(-1, -1, -1, 0); // lazy_window: this triple's rows' predecessors in the launch
    if (WIN) { // rows an earlier batch of this launch rewrites: not asked for before that batch's stamp is there
        if (valid)
            dp = win.deps[(int64_t)blockIdx.y * win.B + slot]; // (win is never written: it stays in kernel-argument memory)
        // Every row is asked for at once, BEFORE the wait: most of a triple's rows have no predecessor in the launch (the user's,
        // the negative item's), and their trip to HBM then runs beside the earlier batches' tails instead of behind them; the
        // rows that do have one are asked for again once its stamp is there (they come from where the finisher wrote them
        // through, not from HBM).  A lane that wants nothing from a load reads the item table's first bytes.
        const bool has_m = h.opt != CREC_OPT_SGD, has_v = h.opt == 0;
        const bool wi = on && (is || blk_whole || DEFER), wj = on && (js || DEFER); // (deferred: every row is brought up to date before it is used)
        const bool want[9] = {on, on, on, on && has_m, on && has_v, wi && has_m, wi && has_v, wj && has_m, wj && has_v};
        const float *const addr[9] = {t.P + ou, t.Q + oi, t.Q + oj, t.mP + ou, t.vP + ou, t.mQ + oi, t.vQ + oi, t.mQ + oj, t.vQ + oj};
        float4 rows[9];
        if (win_sc1) {
            const float *const a1[9] = {want[0] ? addr[0] : t.Q, want[1] ? addr[1] : t.Q, want[2] ? addr[2] : t.Q,
                                        want[3] ? addr[3] : t.Q, want[4] ? addr[4] : t.Q, want[5] ? addr[5] : t.Q,
                                        want[6] ? addr[6] : t.Q, want[7] ? addr[7] : t.Q, want[8] ? addr[8] : t.Q};
            win_load9(a1, rows);
            if (threadIdx.x == 0 && (blockIdx.x & 15) == 0)
                win_t(blockIdx.y, 4, false);
        }
        const bool any_dep = __builtin_amdgcn_ballot_w64(valid && (dp.x >= 0 || dp.y >= 0 || dp.z >= 0)) != 0;
        if (!(win.abl & 1) && any_dep) {
            unsigned spins = 0;
            for (;;) {
                bool pend = false;
                if (valid && l == 0) {
                    const int deps3[3] = {dp.x, dp.y, dp.z};
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        if (deps3[r] >= 0) {
                            const int wbp = deps3[r] >> 20;
                            const int32_t *f = win.done + (int64_t)wbp * 3 * win.B + (r ? win.B : 0) + (deps3[r] & 0xfffff);
                            pend |= __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != win.stamp0 + wbp;
                        }
                    }
                }
                if (__builtin_amdgcn_ballot_w64(pend) == 0)
                    break;
                if (spins < 16) // (a predecessor's stamp is usually 5 - 15 us away, and every hop of a popular row's way from batch to batch
                    __builtin_amdgcn_s_sleep(2); // begins with a poll that sees it: asked again every ~0.5 us, far less often only after 0.5 ms)
                else if (spins < 1024)
                    __builtin_amdgcn_s_sleep(12);
                else
                    for (int zz = 0; zz < 16; ++zz)
                        __builtin_amdgcn_s_sleep(127);
                if ((++spins & 255u) == 0 && (spins >= (1u << 20) || __hip_atomic_load(win.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0)) {
                    if (lane == 0)
                        __hip_atomic_store(win.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
        // the row loads below are relaxed atomics like the polls above: nothing but this keeps the compiler from asking for a row
        // BEFORE its stamp has been seen (the wave issues in program order, and the polls' values are consumed: the hardware keeps it)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (any_dep && threadIdx.x == 0 && (blockIdx.x & 15) == 0) {
            win_t(blockIdx.y, 2, true);
            win_t(blockIdx.y, 3, false);
        }
        if (win_sc1) {
            // the rows with a predecessor, as that batch left them: row by row (the positive item's is the usual one), each only where
            // some triple of the wave has one
            const int dep3[3] = {dp.x, dp.y, dp.z};
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const bool again = valid && dep3[r] >= 0;
                if (__builtin_amdgcn_ballot_w64(again) == 0)
                    continue;
                const int e0 = r, e1 = 3 + 2 * r, e2 = 4 + 2 * r; // the row, its m, its v
                float4 f0, f1, f2;
                win_load3(want[e0] && again ? addr[e0] : t.Q, want[e1] && again ? addr[e1] : t.Q, want[e2] && again ? addr[e2] : t.Q, f0, f1, f2);
                if (again)
                    rows[e0] = f0, rows[e1] = f1, rows[e2] = f2;
            }
            p = want[0] ? rows[0] : zero, qi = want[1] ? rows[1] : zero, qj = want[2] ? rows[2] : zero;
            mp = want[3] ? rows[3] : zero, vp = want[4] ? rows[4] : zero, mi = want[5] ? rows[5] : zero;
            vi = want[6] ? rows[6] : zero, mj = want[7] ? rows[7] : zero, vj = want[8] ? rows[8] : zero;
        }
    }
#if CREC_LAZY_HEADSTART
    // A head start for the rows with a tail.  All waves of the launch are resident at once and ask for their rows in the
    // first microsecond; the memory system then needs the whole body's time to serve them, in no particular order, so a
    // popular row's finisher -- which needs ALL of the row's pieces published -- starts when the body ends, and its
    // publish -> ticket -> sum chain is added on top.  Waves none of whose triples takes a ticket therefore wait a
    // little before asking: the others' rows come back first, and their chains run beside the body instead of after it.
    // (Only where there is such a front to speak of: S-big 16.46 -> 16.11 us with 6 x 0.43 us, 16.7 / 17.9 with 12 / 16.)
    if (!WIN && !front_block && n_slots >= 2048 && 4 * n_blk_slots >= n_slots && __builtin_amdgcn_ballot_w64(valid && (!is || !js)) == 0)
        for (int k = 0; k < CREC_LAZY_HEADSTART; ++k)
            __builtin_amdgcn_s_sleep(16); // 16 x 64 clocks = 0.43 us
#endif
    if (win_sc1) {
        // (asked for above, around the wait for the predecessors' stamps)
    } else if (on) { // every address is known: one round of loads
        p = ld4(t.P + ou);
        qi = ld4(t.Q + oi);
        qj = ld4(t.Q + oj);
        const bool has_m = h.opt != CREC_OPT_SGD, has_v = h.opt == 0; // the slots this optimizer keeps
        // the user's slots are asked for without waiting for its descriptor (is the user row single?): wasted only for users
        // that occur more than once.  (Deferred mode: every row is brought up to date before its values are used.)
        if (has_m)
            mp = ld4s(t.mP + ou);
        if (has_v)
            vp = ld4s(t.vP + ou);
        if ((is || DEFER || blk_whole) && item_adam && has_m)
            mi = ld4s(t.mQ + oi);
        if ((is || DEFER || blk_whole) && item_adam && has_v)
            vi = ld4s(t.vQ + oi);
        if ((js || DEFER) && item_adam && has_m)
            mj = ld4s(t.mQ + oj);
        if ((js || DEFER) && item_adam && has_v)
            vj = ld4s(t.vQ + oj);
    }
    const bool us = du.x < 0; // the user row occurs once in the batch
    float4 mi_raw = mi, vi_raw = vi; // deferred mode: a finisher replays the stored slots itself
    if (DEFER) { // the steps these rows sat out, replayed: from here on p, q_i, q_j are what TF1's tables hold after step - 1
        // (lazy_window: a row's stamp may have been left by an earlier batch of this launch on another XCD -- read where the XCDs meet,
        // behind the wait for that batch's mark above; written through below, in front of this batch's own mark)
        const auto last_of = [&](const int32_t *p_) { return WIN ? __hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p_; };
        const int lu = valid ? last_of(h.lastP + pu.y) : h.step, li = valid ? last_of(h.lastQ + (di.x & 0x7fffffff)) : h.step,
                  lj = valid ? last_of(h.lastQ + (dj.x & 0x7fffffff)) : h.step;
        adam_replay(p, mp, vp, lu, h.step - 1, h);
        adam_replay(qi, mi, vi, li, h.step - 1, h);
        adam_replay(qj, mj, vj, lj, h.step - 1, h);
        if (on && l == 0) { // the single-occurrence rows are finished below
            const auto set_last = [&](int32_t *p_) {
                if (WIN)
                    __hip_atomic_store(p_, h.step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else
                    *p_ = h.step;
            };
            if (us)
                set_last(h.lastP + (du.x & 0x7fffffff));
            if (is)
                set_last(h.lastQ + (di.x & 0x7fffffff));
            if (js)
                set_last(h.lastQ + (dj.x & 0x7fffffff));
        }
    }
    float x;
    const float s = triple_s<LPR>(dot4(p, qi) - dot4(p, qj), &x);
    const float l2 = (0.5f * group_sum<LPR>(dot4(p, p)) + 0.5f * group_sum<LPR>(dot4(qi, qi))) + 0.5f * group_sum<LPR>(dot4(qj, qj));
    if (valid && l == 0 && !(CREC_LAZY_ABL & 16)) {
        xb[pos] = x; // softplus(-x) is taken in bpr_reduce_loss
        l2b[pos] = l2;
    }
    // user row: -s*(q_i - q_j) + reg*p; item rows: -/+ s*p + reg*q.  A row that occurs once is finished here; of a row
    // that occurs more often only the contribution (without the reg term) is published, at its slot's sorted position
    const float ns = -s;
    const float4 cu = make_float4(ns * (qi.x - qj.x), ns * (qi.y - qj.y), ns * (qi.z - qj.z), ns * (qi.w - qj.w));
    const float4 sp = make_float4(s * p.x, s * p.y, s * p.z, s * p.w), nsp = make_float4(-sp.x, -sp.y, -sp.z, -sp.w);

    // 0. front region: the contributions of a row's positive slots (l2 terms included) meet in LDS; the piece's leader adds
    // them in slot (= batch) order and carries the sum from here on
    float4 blk_sum = zero;
    if (front_block && !(CREC_LAZY_ABL & 4)) {
        blk_rows[gw * LPR + l] = (front && act) ? make_float4(nsp.x + h.reg * qi.x, nsp.y + h.reg * qi.y, nsp.z + h.reg * qi.z, nsp.w + h.reg * qi.w)
                                                : zero;
        __syncthreads();
        for (int e = 0; e < blk_n; ++e)
            add4(blk_sum, blk_rows[(gw + e) * LPR + l]);
    }
    // 1. the contribution rows of the multi-occurrence roles, first: their completion is what the tickets wait for
    // lazy_window: a front row all of whose slots are positive ones goes without tickets (lazy_poll_front)
    const bool i_poll = WIN && !DEFER && z.wpart != nullptr && !h.dQ && blk_lead && !blk_whole && i_cnt == i_ci; // (compile-time off outside window launches)
    const bool i_poll_last = i_poll && i_posr + blk_n == i_ci; // the piece that reaches the row's end: its leader finishes the row
    const bool i_publish = !is && !i_poll && (!front || (blk_lead && !blk_whole)); // front region: only a piece's leader speaks for the row
    if (on && !(CREC_LAZY_ABL & 2)) {
        if (!us)
            st4_pub(z.ucontrib, (int64_t)lazy_desc_k(du) * d + 4 * l, cu);
        if (i_publish)
            st4_pub(z.qcontrib, (int64_t)i_k * d + 4 * l, front ? blk_sum : nsp);
        if (i_poll && !i_poll_last)
            st4_pub(z.wpart, (int64_t)i_k * d + 4 * l, blk_sum);
        if (!js)
            st4_pub(z.qcontrib, (int64_t)lazy_desc_k(dj) * d + 4 * l, sp);
    }
    // 2. the tickets of all three roles at once (one round trip, not three).  The contribution rows travel with
    // agent-scope stores; the wave waits for their completion first -- the same guarantee as a release fence for these
    // buffers, without writing back the whole L2 (the row loads above have long been consumed: only these stores are
    // outstanding)
    const bool any_multi = !(CREC_LAZY_ABL & 2) && __builtin_amdgcn_ballot_w64(valid && (!us || i_publish || !js)) != 0; // false for the usual wave on large tables
    const int wi = front ? blk_n : 1;
    int tku = -1, tki = -1, tkj = -1;
    if (any_multi) {
        pub_release();
        if (valid && l == 0) {
            const LazyTicket qu = lazy_ticket_of(du, z.CH), qi_ = lazy_ticket_of(di, z.CH), qj_ = lazy_ticket_of(dj, z.CH);
            if (!us)
                tku = __hip_atomic_fetch_add((qu.chunked ? z.uchk_t : z.urow_t) + qu.idx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (i_publish)
                tki = __hip_atomic_fetch_add((qi_.chunked ? z.qchk_t : z.qrow_t) + qi_.idx, wi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!js)
                tkj = __hip_atomic_fetch_add((qj_.chunked ? z.qchk_t : z.qrow_t) + qj_.idx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // 3. the rows that occur once are finished here, while the tickets travel
    if (on) {
        if (us) {
            const float4 gp = make_float4(cu.x + h.reg * p.x, cu.y + h.reg * p.y, cu.z + h.reg * p.z, cu.w + h.reg * p.w);
            float4 np = p;
            opt4(np, mp, vp, gp, h);
            if (win_wt) {
                st4_wtg(t.P + ou, np);
                if (h.opt != CREC_OPT_SGD)
                    st4_wtg(t.mP + ou, mp);
                if (h.opt == 0)
                    st4_wtg(t.vP + ou, vp);
            } else {
                st4s(t.P + ou, np);
                if (h.opt != CREC_OPT_SGD)
                    st4s(t.mP + ou, mp);
                if (h.opt == 0)
                    st4s(t.vP + ou, vp);
            }
        }
        if (is) {
            const float4 gi = make_float4(nsp.x + h.reg * qi.x, nsp.y + h.reg * qi.y, nsp.z + h.reg * qi.z, nsp.w + h.reg * qi.w);
            if (item_adam) {
                float4 nq = qi;
                opt4(nq, mi, vi, gi, h);
                if (win_wt) {
                    st4_wtg(t.Q + oi, nq);
                    if (h.opt != CREC_OPT_SGD)
                        st4_wtg(t.mQ + oi, mi);
                    if (h.opt == 0)
                        st4_wtg(t.vQ + oi, vi);
                } else {
                    st4s(t.Q + oi, nq);
                    if (h.opt != CREC_OPT_SGD)
                        st4s(t.mQ + oi, mi);
                    if (h.opt == 0)
                        st4s(t.vQ + oi, vi);
                }
            } else {
                st4_dq(h.dQ, oi, gi, h.dq_sys, true);
            }
        }
        if (js) {
            const float4 gj = make_float4(sp.x + h.reg * qj.x, sp.y + h.reg * qj.y, sp.z + h.reg * qj.z, sp.w + h.reg * qj.w);
            if (item_adam) {
                float4 nq = qj;
                opt4(nq, mj, vj, gj, h);
                if (win_wt) {
                    st4_wtg(t.Q + oj, nq);
                    if (h.opt != CREC_OPT_SGD)
                        st4_wtg(t.mQ + oj, mj);
                    if (h.opt == 0)
                        st4_wtg(t.vQ + oj, vj);
                } else {
                    st4s(t.Q + oj, nq);
                    if (h.opt != CREC_OPT_SGD)
                        st4s(t.mQ + oj, mj);
                    if (h.opt == 0)
                        st4s(t.vQ + oj, vj);
                }
            } else {
                st4_dq(h.dQ, oj, gj, h.dq_sys, true);
            }
        }
    }
    if (win_wt) { // the rows finished above have arrived: this batch's stamps for their next readers in the window
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (valid && l == 0) {
            if (us)
                __hip_atomic_store(h.win_done_u + lazy_desc_start(du), h.win_stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (is)
                __hip_atomic_store(h.win_done_q + lazy_desc_start(di), h.win_stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (js)
                __hip_atomic_store(h.win_done_q + lazy_desc_start(dj), h.win_stamp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (blk_whole) // a front row that lies inside this workgroup: finished from LDS, nothing published
        lazy_finish_row<LPR, false, DEFER>(t, d, act, di.x & 0x7fffffff, qi, blk_sum, mi_raw, vi_raw, h, lazy_desc_start(di));
    if (WIN && i_poll_last)
        lazy_poll_front<LPR, DEFER>(t, d, act, di, qi, blk_sum, z.wpart, SPW, h, z.err);
    if (!any_multi || (CREC_LAZY_ABL & 1))
        return;
    // 4. whoever completes a row's (chunk's) counter finishes it
    const int lead = lane - l;
    tku = __shfl(tku, lead, 64);
    tki = __shfl(tki, lead, 64);
    tkj = __shfl(tkj, lead, 64);
    if (valid && !us)
        lazy_role_tail<LPR, true, DEFER, WIN>(t, d, act, du, tku, 1, p, z.ucontrib, z.upart, z.urow_t, z.uchk_t, z.CH, SPW, h);
    if (valid && i_publish)
        lazy_role_tail<LPR, false, DEFER, WIN>(t, d, act, di, tki, wi, qi, z.qcontrib, z.qpart, z.qrow_t, z.qchk_t, z.CH, SPW, h);
    if (valid && !js)
        lazy_role_tail<LPR, false, DEFER, WIN>(t, d, act, dj, tkj, 1, qj, z.qcontrib, z.qpart, z.qrow_t, z.qchk_t, z.CH, SPW, h);
    if (WIN && threadIdx.x == 0 && (blockIdx.x & 15) == 0)
        win_t(blockIdx.y, 1, false);
}

// the tables of the second buffer set back into the caller's (an odd number of tf1 launches): up to six copies in one launch
// (rows are multiples of four floats, the buffers 16-byte aligned)
struct CopyTables {
    float4 *dst[6];
    const float4 *src[6];
    int64_t n4[6];
};
// table k's share of workgroup bx of nbx (any number of threads per workgroup)
__device__ __forceinline__ void copy_tables_block(const CopyTables &c, int k, int bx, int nbx)
{
    float4 *__restrict__ dst = c.dst[k];
    const float4 *__restrict__ src = c.src[k];
    const int64_t n = c.n4[k], nt = blockDim.x;
    for (int64_t o = (int64_t)bx * nt + threadIdx.x; o < n; o += (int64_t)nbx * nt)
        dst[o] = src[o];
}
__global__ __launch_bounds__(256) void copy_tables(const CopyTables c) { copy_tables_block(c, blockIdx.y, blockIdx.x, gridDim.x); }

// deferred mode: every row of a table brought up to the state after step h.step (before evaluation / a checkpoint);
// a lane group per row
__global__ __launch_bounds__(256) void bpr_adam_flush(float *__restrict__ var, float *__restrict__ m, float *__restrict__ v,
                                                      int32_t *__restrict__ last, int64_t rows, int d, int lpr, Hyper h)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = t / lpr;
    const int l = (int)(t % lpr);
    if (r >= rows || 4 * l >= d)
        return;
    const int L = last[r];
    if (L >= h.step)
        return;
    const int64_t ro = r * d + 4 * l;
    float4 a = ld4(var + ro), b = ld4(m + ro), c = ld4(v + ro);
    adam_replay(a, b, c, L, h.step, h);
    st4(var + ro, a);
    st4(m + ro, b);
    st4(v + ro, c);
    // the row's lanes read `last` before any of them may write it: the writer is the lane that finishes the row's last slice
    if (4 * (l + 1) >= d)
        last[r] = h.step;
}

// ------------------------------------------------------------------------------------
// bpr_adam_dense (data-parallel step): TF1 Adam of every row of Q from the all-reduced gradient.
// A row no rank touched has g == 0, which is exactly the untouched sweep (m*b1 + 0 == m*b1).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bpr_adam_dense(float *__restrict__ var, float *__restrict__ m,
                                                      float *__restrict__ v, const float *__restrict__ g, int64_t n4,
                                                      Hyper h)
{
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += (int64_t)gridDim.x * blockDim.x) {
        float4 a = ld4(var + 4 * k), b = ld4(m + 4 * k), c = ld4(v + 4 * k);
        adam4(a, b, c, ld4(g + 4 * k), true, h);
        st4s(var + 4 * k, a);
        st4s(m + 4 * k, b);
        st4s(v + 4 * k, c);
    }
}

// ------------------------------------------------------------------------------------
// bpr_reduce_loss: RL_PARTS workgroups per batch, fixed summation order, float64 accumulation.
//   dense: rowterm[0 .. n_dense).  lazy (n_dense == 0): one l2 term per triple, rowterm[0 .. nb).
// Part p of a batch sums the p-th quarter of the x_b terms (softplus) and of the row terms; the part that draws the batch's last
// ticket adds the four partial sums in part order.  (One workgroup per batch -- 6 144 expf + log1pf on ONE CU -- took 10-17 us
// whatever the number of batches: a tenth of a 20-batch call.)  parts: [batches][RL_PARTS] doubles, tickets: [batches], zero
// between launches.
// ------------------------------------------------------------------------------------
constexpr int RL_THREADS = 512, RL_PARTS = 4;
struct ReduceLoss {
    const float *ws;
    int64_t stride, n_total;
    int32_t B;
    int64_t n_dense;
    float reg;
    double *parts;
    int32_t *tickets;
    float *loss_per_batch;
};
__device__ __forceinline__ void reduce_loss_block(const ReduceLoss &a, int blk, double *part)
{
    const float *__restrict__ ws = a.ws;
    const int64_t stride = a.stride, n_total = a.n_total, n_dense = a.n_dense;
    const int32_t B = a.B;
    const float reg = a.reg;
    double *__restrict__ parts = a.parts;
    int32_t *__restrict__ tickets = a.tickets;
    float *__restrict__ loss_per_batch = a.loss_per_batch;
    const int batch = blk / RL_PARTS, p = blk % RL_PARTS;
    const int64_t t0 = (int64_t)batch * B;
    const int nb = (int)((n_total - t0) < B ? (n_total - t0) : B);
    const float *xb = ws + batch * stride, *rowterm = xb + B;
    // a thread's terms are added in ascending k; the loads of eight of them are issued together
    const auto sum_f = [](const float *q, int64_t n, auto f, double acc) {
        for (int64_t k = threadIdx.x; k < n; k += 8 * RL_THREADS) {
            float t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                t[e] = k + e * RL_THREADS < n ? q[k + e * RL_THREADS] : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (k + e * RL_THREADS < n)
                    acc += (double)f(t[e]);
        }
        return acc;
    };
    const auto share = [&](int64_t n, int64_t *lo) { // this part's [lo, lo + count) of n terms
        const int64_t per = (n + RL_PARTS - 1) / RL_PARTS;
        *lo = per * p < n ? per * p : n;
        return (*lo + per) < n ? per : n - *lo;
    };
    const auto ident = [](float x) { return x; };
    int64_t lo_x, lo_r;
    const int64_t nx = share(nb, &lo_x), nr = share(n_dense > 0 ? n_dense : (int64_t)nb, &lo_r); // lazy mode: one l2 term per triple
    const double s = sum_f(xb + lo_x, nx, [](float x) { return tf_softplus(-x); }, 0.0);
    const double l2 = sum_f(rowterm + lo_r, nr, ident, 0.0);
    part[threadIdx.x] = s + (double)reg * l2;
    __syncthreads();
    for (int off = RL_THREADS / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(parts + (int64_t)batch * RL_PARTS + p, part[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the agent-scope store has arrived where the other workgroups read)
        if (__hip_atomic_fetch_add(tickets + batch, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == RL_PARTS - 1) {
            double tot = 0.0;
            for (int q = 0; q < RL_PARTS; ++q) // part order: the sum does not depend on which part came last
                tot += __hip_atomic_load(parts + (int64_t)batch * RL_PARTS + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            loss_per_batch[batch] = (float)tot;
            __hip_atomic_store(tickets + batch, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch
        }
    }
}
__global__ __launch_bounds__(RL_THREADS) void bpr_reduce_loss(const float *__restrict__ ws, int64_t stride, int64_t n_total,
                                                       int32_t B, int64_t n_dense, const int32_t *__restrict__ n_u,
                                                       const int32_t *__restrict__ n_q, float reg, double *__restrict__ parts,
                                                       int32_t *__restrict__ tickets, float *__restrict__ loss_per_batch)
{
    __shared__ double part[RL_THREADS];
    reduce_loss_block(ReduceLoss{ws, stride, n_total, B, n_dense, reg, parts, tickets, loss_per_batch}, blockIdx.x, part);
}
// The tail of a call that ends in the second buffer set: the last chunk's loss reduction and the tables' way home are
// independent of each other and both wait for the last step -- ONE launch (the reduction's workgroups first, then
// TAIL_COPY_BX workgroups per table) instead of two: a launch boundary less behind a 20-batch call.
constexpr int TAIL_COPY_BX = 32;
__global__ __launch_bounds__(RL_THREADS) void bpr_call_tail(const ReduceLoss rl, int n_loss_blocks, const CopyTables c)
{
    __shared__ double part[RL_THREADS];
    if ((int)blockIdx.x < n_loss_blocks) {
        reduce_loss_block(rl, blockIdx.x, part);
        return;
    }
    const int cb = (int)blockIdx.x - n_loss_blocks;
    copy_tables_block(c, cb / TAIL_COPY_BX, cb % TAIL_COPY_BX, TAIL_COPY_BX);
}

#include "shard_kernels.h"

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
struct StepArgs {
    crec_ctx *ctx;
    Tables cur, other; // tf1: read `cur`, write `other`.  lazy: `cur` in place.
    int d;
    int64_t U, I;
    int32_t B;
    float *contrib, *xb, *rowterm, *partial;
    int32_t *tickets;
    Hyper h;
    hipStream_t st;
    const int32_t *nwide; // host copy of the chunk's wide-group counts [nbc][2] (tf1 mode)
    // Workgroups are dealt round-robin to the 8 XCDs, and a table row is cheap to read on the XCD that wrote it in
    // the previous step.  The workgroups in front of the row waves (wide groups, heavy-row chunk waves) change in
    // number from batch to batch; if that shifted the row waves by anything but a multiple of 8 workgroups, every
    // row would move to another XCD every step: the same kernel and work measured 9.5 us per launch so against 7.1
    // (CREC_WIDE_ROUND=7 vs 8).  Every step of a chunk is launched with the chunk's largest counts; surplus waves
    // exit at once.
    int pad_wide, pad_chunks_u, pad_chunks_q; // max over the chunk of NWu+NWq, HIu, HIq
    LazyBufs lazy;                            // lazy mode: contribution rows, partials, ticket counters
    int32_t nb_cur;                           // triples in the batch being launched
    int wide_round;                           // CREC_WIDE_ROUND, read once per epoch (0: unset)
};

// hn: this batch's {user heavy rows, user chunks, item heavy rows, item chunks}, read back by the host
// once per chunk of batches, so grids are sized to the real counts and the finish launch is skipped
// when a batch has no heavy row at all.
HeavyArgs heavy_args(const StepArgs &a, const IndexWs &w, int64_t b, const int32_t *hn)
{
    HeavyArgs hv;
    hv.hn = w.hn + b * HN;
    hv.u_hitems = w.u_hitems + b * w.HIu;
    hv.q_hitems = w.q_hitems + b * w.HIq;
    hv.u_hrows = w.u_hrows + b * w.HRu;
    hv.q_hrows = w.q_hrows + b * w.HRq;
    hv.partial = a.partial;
    hv.tickets = a.tickets;
    hv.HRu_max = (int)w.HRu;
    hv.u_wide = w.u_wide + b * w.WMu;
    hv.q_wide = w.q_wide + b * w.WMq;
    hv.NWu = hv.NWq = hv.NWpad = 0; // launch_fused fills them in
    hv.HRu = hn[0];
    hv.HIu = hn[1];
    hv.HRq = hn[2];
    hv.HIq = hn[3];
    return hv;
}

template <int LPR, bool FULL> void launch_fused(const StepArgs &a, const IndexWs &w, int64_t b, const int32_t *hn)
{
    constexpr int R = Geo<LPR, ROWS_PER_WAVE>::R;
    HeavyArgs hv = heavy_args(a, w, b, hn);
    if (w.wide_rows && a.nwide) {
        hv.NWu = a.nwide[b * 2 + 0];
        hv.NWq = a.nwide[b * 2 + 1];
        hv.NWpad = a.pad_wide;
        if (a.wide_round > 0) // experiment: this batch's own count rounded up to a multiple
            hv.NWpad = (hv.NWu + hv.NWq + a.wide_round - 1) / a.wide_round * a.wide_round;
    }
    const int64_t waves = a.pad_chunks_u + a.pad_chunks_q + crec_ceil_div(a.U, R) + crec_ceil_div(a.I, R);
    crec_prof_begin(a.ctx, CREC_K_STEP_FUSED, a.st);
    CREC_KLAUNCH(a.ctx, (bpr_step_fused<LPR, FULL>), dim3((unsigned)(hv.NWpad + crec_ceil_div(waves, CREC_FUSED_BLOCK / 64))),
                 dim3(CREC_FUSED_BLOCK), 0, a.st, a.cur, a.other,
                       a.d, a.U, a.I, w.u_rowseg + b * w.U, w.u_occ + b * w.B, w.q_rowseg + b * w.I,
                       w.q_occ + b * 2 * w.B, hv, a.xb, a.rowterm, a.h);
    crec_prof_end(a.ctx, a.st);
}

template <int LPR, bool FULL, bool BIG> void launch_lazy_geo(const StepArgs &a, const IndexWs &w, int64_t b)
{
    constexpr int SPW = LazyGeo<LPR, BIG>::SPW, THREADS = LazyGeo<LPR, BIG>::THREADS;
    const dim3 grid((unsigned)crec_ceil_div((int64_t)a.nb_cur, SPW)), block(THREADS);
    const size_t lds = sizeof(float4) * THREADS;
    crec_prof_begin(a.ctx, CREC_K_USER_SIDE, a.st);
    if (a.h.lr_tab)
        CREC_KLAUNCH(a.ctx, (bpr_triples_lazy<LPR, FULL, true, false, BIG>), grid, block, lds, a.st, a.cur, a.d,
                     (const int2 *)(w.slot_pu + b * w.B), (int)a.nb_cur, (const int32_t *)(w.nslots + b * 2), (const int4 *)(w.u_desc + b * w.B),
                     (const int4 *)(w.i_desc + b * w.B), (const int4 *)(w.j_desc + b * w.B), a.lazy, a.xb, a.rowterm, a.h, LazyWin{});
    else
        CREC_KLAUNCH(a.ctx, (bpr_triples_lazy<LPR, FULL, false, false, BIG>), grid, block, lds, a.st, a.cur, a.d,
                     (const int2 *)(w.slot_pu + b * w.B), (int)a.nb_cur, (const int32_t *)(w.nslots + b * 2), (const int4 *)(w.u_desc + b * w.B),
                     (const int4 *)(w.i_desc + b * w.B), (const int4 *)(w.j_desc + b * w.B), a.lazy, a.xb, a.rowterm, a.h, LazyWin{});
    crec_prof_end(a.ctx, a.st);
}
// (the index was laid out for w.spw slots per workgroup: the kernel of that geometry)
template <int LPR, bool FULL> void launch_lazy(const StepArgs &a, const IndexWs &w, int64_t b, const int32_t *hn)
{
    (void)hn;
    if constexpr (LazyGeo<LPR, true>::SPW != LazyGeo<LPR, false>::SPW) {
        if (w.spw == LazyGeo<LPR, true>::SPW)
            return launch_lazy_geo<LPR, FULL, true>(a, w, b);
    }
    launch_lazy_geo<LPR, FULL, false>(a, w, b);
}

// lazy_window: the batches [b, b + win.W) of the chunk in one launch (blockIdx.y = the batch)
template <int LPR, bool FULL, bool BIG> void launch_lazy_window_geo(const StepArgs &a, const IndexWs &w, int64_t b, const LazyWin &win)
{
    constexpr int SPW = LazyGeo<LPR, BIG>::SPW, THREADS = LazyGeo<LPR, BIG>::THREADS;
    const dim3 grid((unsigned)crec_ceil_div((int64_t)a.B, SPW), (unsigned)win.W), block(THREADS);
    const size_t lds = sizeof(float4) * THREADS;
    a.ctx->prof.next_units = win.W;
    crec_prof_begin(a.ctx, CREC_K_USER_SIDE, a.st);
    if (a.h.lr_tab)
        CREC_KLAUNCH(a.ctx, (bpr_triples_lazy<LPR, FULL, true, true, BIG>), grid, block, lds, a.st, a.cur, a.d, (const int2 *)(w.slot_pu + b * w.B), 0,
                     (const int32_t *)(w.nslots + b * 2), (const int4 *)(w.u_desc + b * w.B), (const int4 *)(w.i_desc + b * w.B),
                     (const int4 *)(w.j_desc + b * w.B), a.lazy, a.xb, a.rowterm, a.h, win);
    else
        CREC_KLAUNCH(a.ctx, (bpr_triples_lazy<LPR, FULL, false, true, BIG>), grid, block, lds, a.st, a.cur, a.d, (const int2 *)(w.slot_pu + b * w.B), 0,
                 (const int32_t *)(w.nslots + b * 2), (const int4 *)(w.u_desc + b * w.B), (const int4 *)(w.i_desc + b * w.B),
                 (const int4 *)(w.j_desc + b * w.B), a.lazy, a.xb, a.rowterm, a.h, win);
    crec_prof_end(a.ctx, a.st);
    a.ctx->prof.next_units = 1;
}
template <int LPR, bool FULL> void launch_lazy_window(const StepArgs &a, const IndexWs &w, int64_t b, const LazyWin &win)
{
    if constexpr (LazyGeo<LPR, true>::SPW != LazyGeo<LPR, false>::SPW) {
        if (w.spw == LazyGeo<LPR, true>::SPW)
            return launch_lazy_window_geo<LPR, FULL, true>(a, w, b, win);
    }
    launch_lazy_window_geo<LPR, FULL, false>(a, w, b, win);
}
using WinFn = void (*)(const StepArgs &, const IndexWs &, int64_t, const LazyWin &);
WinFn pick_window(int d)
{
    int lpr = 1;
    while (lpr * 4 < d)
        lpr <<= 1;
    const bool full = d == 4 * lpr;
#define CREC_PICK(L)                                                                 \
    case L:                                                                          \
        return full ? launch_lazy_window<L, true> : launch_lazy_window<L, false>;
    switch (lpr) {
        CREC_PICK(1)
        CREC_PICK(2)
        CREC_PICK(4)
        CREC_PICK(8)
        CREC_PICK(16)
        CREC_PICK(32)
        CREC_PICK(64)
    }
#undef CREC_PICK
    return nullptr;
}

// lazy mode: contribution rows and partials by sorted slot position (user side B slots, item side 2B), then the
// ticket counters, cleared here once -- the kernels leave them at zero
int setup_lazy_bufs(crec_ctx *ctx, int64_t B, int d, int CH, hipStream_t st, LazyBufs *z, int nwin = 1)
{
    // (lazy_window: nwin batches' buffers back to back, 6 B rows and 6 B counters each; z describes the first batch's)
    const size_t rows = 6 * (size_t)B * (size_t)nwin, row_bytes = align_up(sizeof(float) * rows * d, 256), ctr_bytes = sizeof(int32_t) * rows;
    if (int rc = crec_reserve(ctx, &ctx->contrib, row_bytes + ctr_bytes))
        return rc;
    float *f = (float *)ctx->contrib.ptr;
    z->ucontrib = f;
    z->qcontrib = f + (size_t)B * d;
    z->upart = f + 3 * (size_t)B * d;
    z->qpart = f + 4 * (size_t)B * d;
    int32_t *c = (int32_t *)((char *)ctx->contrib.ptr + row_bytes);
    z->urow_t = c;
    z->uchk_t = c + B;
    z->qrow_t = c + 2 * B;
    z->qchk_t = c + 4 * B;
    z->CH = CH;
    z->wpart = nullptr, z->err = nullptr; // (set by the callers that use lazy_poll_front)
    if (ctx->lazy_tickets_at != (char *)c || ctx->lazy_tickets_bytes != ctr_bytes) {
        CREC_HIP(ctx, hipMemsetAsync(c, 0, ctr_bytes, st));
        ctx->lazy_tickets_at = (char *)c;
        ctx->lazy_tickets_bytes = ctr_bytes;
    }
    return CREC_OK;
}

using StepFn = void (*)(const StepArgs &, const IndexWs &, int64_t, const int32_t *);

// bpr_reduce_loss's scratch for `nbc` batches: RL_PARTS partial sums and one ticket per batch (the tickets are cleared when the buffer
// moves or grows; the kernel leaves them at zero)
int loss_parts(crec_ctx *ctx, int64_t nbc, hipStream_t st, double **parts, int32_t **tickets)
{
    const size_t pbytes = align_up(sizeof(double) * RL_PARTS * (size_t)nbc, 256), tbytes = sizeof(int32_t) * (size_t)nbc;
    const void *before = ctx->loss_parts.ptr;
    const size_t had = ctx->loss_parts.bytes;
    if (int rc = crec_reserve(ctx, &ctx->loss_parts, pbytes + tbytes + 4096))
        return rc;
    if (ctx->loss_parts.ptr != before || ctx->loss_parts.bytes != had) // (a fresh buffer)
        CREC_HIP(ctx, hipMemsetAsync(ctx->loss_parts.ptr, 0, ctx->loss_parts.bytes, st));
    // (laid out by what the buffer holds, so that the tickets stay where they were cleared while it does not move)
    const size_t cap = (ctx->loss_parts.bytes - 256) / (sizeof(double) * RL_PARTS + sizeof(int32_t)); // batches it holds (>= nbc)
    *parts = (double *)ctx->loss_parts.ptr;
    *tickets = (int32_t *)((char *)ctx->loss_parts.ptr + align_up(sizeof(double) * RL_PARTS * cap, 256));
    return CREC_OK;
}

// rows of one side dealt to nwg workgroups of rpb slots: pos_of[row] = workgroup * rpb + slot, row_of[position] = row or -1
void flow_deal_host(const int32_t *cnt, int n, int nwg, int rpb, int32_t *pos_of, int32_t *row_of)
{
    std::vector<int32_t> order((size_t)n);
    for (int r = 0; r < n; ++r)
        order[(size_t)r] = r;
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return cnt[a] > cnt[b]; });
    for (int64_t p_ = 0; p_ < (int64_t)nwg * rpb; ++p_)
        row_of[p_] = -1;
    using Entry = std::pair<int64_t, int32_t>; // {work so far, workgroup}: the least-loaded one with a free slot takes the next row
    std::priority_queue<Entry, std::vector<Entry>, std::greater<Entry>> heap;
    std::vector<int32_t> used((size_t)nwg, 0);
    for (int w_ = 0; w_ < nwg; ++w_)
        heap.push({0, w_});
    for (int32_t row : order) {
        const Entry e = heap.top();
        heap.pop();
        const int32_t w_ = e.second, slot = used[(size_t)w_]++;
        pos_of[row] = w_ * rpb + slot;
        row_of[(int64_t)w_ * rpb + slot] = row;
        if (used[(size_t)w_] < rpb)
            heap.push({e.first + cnt[row] + 1, w_}); // (+ 1: a row costs its owner a little even when no record names it)
    }
}

// bpr_steps_flow<LPR, FULL>: the launch, and the workgroups of it one CU holds at a time
struct FlowKernel {
    void (*launch)(crec_ctx *, unsigned grid, hipStream_t, const Tables &, const Tables &, int, int64_t, int64_t, const FlowArgs &, const Hyper &);
    const void *fn;
    int rpb_u, rpb_q, lpr_log2, full; // rows per user / item workgroup
};
template <int LPR, bool FULL>
void launch_flow(crec_ctx *ctx, unsigned grid, hipStream_t st, const Tables &src, const Tables &dst, int d, int64_t U, int64_t I,
                 const FlowArgs &f, const Hyper &h)
{
    CREC_KLAUNCH(ctx, (bpr_steps_flow<LPR, FULL>), dim3(grid), dim3(CREC_FUSED_BLOCK), 0, st, src, dst, d, U, I, f, h);
}
FlowKernel pick_flow(int d)
{
    int lpr = 1, lg = 0;
    while (lpr * 4 < d)
        lpr <<= 1, ++lg;
    const bool full = d == 4 * lpr;
#define CREC_PICK(L)                                                                                                              \
    case L:                                                                                                                       \
        return full ? FlowKernel{launch_flow<L, true>, (const void *)bpr_steps_flow<L, true>,                                     \
                                 FlowGeo<L, 1>::RPB, FlowGeo<L, CREC_FLOW_ITEM_DIV>::RPB, lg, 1}                                  \
                    : FlowKernel{launch_flow<L, false>, (const void *)bpr_steps_flow<L, false>,                                   \
                                 FlowGeo<L, 1>::RPB, FlowGeo<L, CREC_FLOW_ITEM_DIV>::RPB, lg, 0};
    switch (lpr) {
        CREC_PICK(1)
        CREC_PICK(2)
        CREC_PICK(4)
        CREC_PICK(8)
        CREC_PICK(16)
        CREC_PICK(32)
        CREC_PICK(64)
    }
#undef CREC_PICK
    return FlowKernel{nullptr, nullptr, 0, 0, 0, 0};
}

StepFn pick_step(int d, bool dense)
{
    int lpr = 1;
    while (lpr * 4 < d)
        lpr <<= 1;
    const bool full = d == 4 * lpr;
#define CREC_PICK(L)                                                                                      \
    case L:                                                                                               \
        return dense ? (full ? launch_fused<L, true> : launch_fused<L, false>)                            \
                     : (full ? launch_lazy<L, true> : launch_lazy<L, false>);
    switch (lpr) {
        CREC_PICK(1)
        CREC_PICK(2)
        CREC_PICK(4)
        CREC_PICK(8)
        CREC_PICK(16)
        CREC_PICK(32)
        CREC_PICK(64)
    }
#undef CREC_PICK
    return nullptr;
}

} // namespace

namespace {
// deferred mode: lr_s of every step so far, kept on the device (TF1's fp32 running beta powers from step 1 on)
int lr_history(crec_ctx *ctx, const crec_adam *opt, int64_t upto, hipStream_t st, const float **dev)
{
    if (ctx->lr_hist.empty() || ctx->lr_hist_key[0] != opt->lr || ctx->lr_hist_key[1] != opt->beta1 || ctx->lr_hist_key[2] != opt->beta2) {
        ctx->lr_hist.assign(1, 0.f); // [0] unused: steps are 1-based
        ctx->lr_hist_key[0] = opt->lr;
        ctx->lr_hist_key[1] = opt->beta1;
        ctx->lr_hist_key[2] = opt->beta2;
        ctx->lr_hist_b1p = opt->beta1;
        ctx->lr_hist_b2p = opt->beta2;
        ctx->lr_hist_uploaded = 0;
    }
    while ((int64_t)ctx->lr_hist.size() <= upto) {
        ctx->lr_hist.push_back((opt->lr * sqrtf(1.0f - ctx->lr_hist_b2p)) / (1.0f - ctx->lr_hist_b1p));
        ctx->lr_hist_b1p = ctx->lr_hist_b1p * opt->beta1;
        ctx->lr_hist_b2p = ctx->lr_hist_b2p * opt->beta2;
    }
    const void *before = ctx->lr_hist_dev.ptr;
    if (int rc = crec_reserve(ctx, &ctx->lr_hist_dev, sizeof(float) * (ctx->lr_hist.size() + 4096)))
        return rc;
    if (ctx->lr_hist_dev.ptr != before)
        ctx->lr_hist_uploaded = 0;
    if (ctx->lr_hist_uploaded < ctx->lr_hist.size()) {
        // a blocking copy from pageable memory: the vector may grow (and move) on the next call
        CREC_HIP(ctx, hipMemcpyAsync((float *)ctx->lr_hist_dev.ptr + ctx->lr_hist_uploaded, ctx->lr_hist.data() + ctx->lr_hist_uploaded,
                                     sizeof(float) * (ctx->lr_hist.size() - ctx->lr_hist_uploaded), hipMemcpyHostToDevice, st));
        CREC_HIP(ctx, hipStreamSynchronize(st));
        ctx->lr_hist_uploaded = ctx->lr_hist.size();
    }
    *dev = (const float *)ctx->lr_hist_dev.ptr;
    return CREC_OK;
}

int train_epoch_impl(crec_ctx *ctx, const crec_bpr_tables *t, crec_adam *opt, const int32_t *u, const int32_t *i,
                     const int32_t *j, int64_t N, int32_t B, float *loss_per_batch, void *stream, bool dp,
                     const crec_bpr_deferred *df = nullptr)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, t && opt, "null argument");
    CREC_REQUIRE(ctx, N >= 0, "negative N");
    CREC_REQUIRE(ctx, N == 0 || (u && i && j && loss_per_batch), "null argument");
    CREC_REQUIRE(ctx, t->P && t->Q && t->mP && t->vP && t->mQ && t->vQ, "null table pointer");
    CREC_REQUIRE(ctx, t->d >= 4 && t->d <= 256 && t->d % 4 == 0, "embed_size %d: need a multiple of 4 in [4, 256]", t->d);
    CREC_REQUIRE(ctx, t->U >= 1 && t->I >= 1 && t->U < ((int64_t)1 << 31) && t->I < ((int64_t)1 << 31), "table rows");
    CREC_REQUIRE(ctx, B >= 1 && B <= LAZY_MAX_B, "batch_size %d: batches of 1..%d triples", B, LAZY_MAX_B);
    CREC_REQUIRE(ctx, B <= MAX_B || opt->adam_mode != CREC_ADAM_TF1,
                 "batch_size %d with adam_mode tf1 and beta1 > 0.9: batches beyond %d run as deferred Adam, which replays idle steps exactly only for "
                 "beta1 <= 0.9 (or the data-parallel step: batches of at most %d); use lazy, SGD or Adagrad", B, MAX_B, MAX_B);
    CREC_REQUIRE(ctx, opt->adam_mode == CREC_ADAM_TF1 || opt->adam_mode == CREC_ADAM_LAZY || opt->adam_mode == CREC_OPT_SGD ||
                          opt->adam_mode == CREC_OPT_ADAGRAD || opt->adam_mode == CREC_ADAM_DEFERRED, "adam_mode");
    CREC_REQUIRE(ctx, (opt->adam_mode == CREC_ADAM_DEFERRED) == (df != nullptr),
                 "adam_mode deferred goes through crec_bpr_train_epoch_deferred (and only it)");
    CREC_REQUIRE(ctx, !df || (df->lastP && df->lastQ), "null last-step arrays");
    const float *tabs[] = {t->P, t->Q, t->mP, t->vP, t->mQ, t->vQ};
    for (const float *p : tabs)
        CREC_REQUIRE(ctx, ((uintptr_t)p & 15) == 0, "table base not 16-byte aligned");
    if (N == 0)
        return CREC_OK;
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const bool dense = opt->adam_mode == CREC_ADAM_TF1;
    crec_xch *xdp = nullptr; // the data-parallel step's all-reduces go through the ctx's transport (RCCL or the peer windows)
    if (dp) {
        CREC_REQUIRE(ctx, ctx->comm != nullptr || ctx->xch != nullptr, "crec_comm_init (or crec_comm_peer_connect) has not been called on this ctx");
        CREC_REQUIRE(ctx, dense, "the data-parallel step is built for adam_mode tf1");
        xdp = crec_xch_of(ctx);
        CREC_REQUIRE(ctx, (int64_t)t->I * t->d * 4 <= xdp->max_bytes(), "the transport's window is too small for the all-reduce of dQ (%lld bytes)",
                     (long long)t->I * t->d * 4);
    }
    StepFn step = pick_step(t->d, dense);
    const int64_t nb = crec_ceil_div(N, B);
    const float *lr_tab = nullptr;
    if (df) {
        CREC_REQUIRE(ctx, opt->step + nb < ((int64_t)1 << 31), "step counter");
        if (int rc = lr_history(ctx, opt, opt->step + nb, st, &lr_tab))
            return rc;
    }
    // occurrences per heavy-row chunk: short chunks shorten the longest wave, every chunk wave pays a ticket on its
    // row's counter.  Item side of S-big: CH 8 -> 12.4 us, 16 -> 11.4, 32 -> 12.0 (with fence-published partials the
    // sweep read 16 -> 16.3, 32 -> 13.7, 64 -> 15.3).  tf1 mode at ml-1m shape: 32 -> 14.4 us/step, 64 -> 13.2.
    const int CH = ctx->knob_ch > 0 ? ctx->knob_ch : (dense ? 64 : 16);
    CREC_REQUIRE(ctx, CH >= 1 && CH <= 255, "option ch: 1..255");

    // wide groups (tf1 mode): the rows of a row wave that hold more than `wide_total` occurrences together go to a
    // whole workgroup.  ctx->knob_wide: 0 turns it off, any other value >= 0 is the threshold (crec_ctx_set_option).
    int wide_rows = 1;
    {
        int lpr = 1;
        while (lpr * 4 < t->d)
            lpr <<= 1;
        wide_rows = 64 / lpr < ROWS_PER_WAVE ? 64 / lpr : ROWS_PER_WAVE;
    }
    int wide_total = (!dense || wide_rows < 2) ? 0 : (ctx->knob_wide >= 0 ? ctx->knob_wide : 16);

    // tf1 mode, all steps of a chunk in ONE launch (bpr_steps_flow) where every wave of it is resident at once and the version
    // histories stay under 1 GiB; else (and for the data-parallel step, whose item rows leave the launch every step) a launch per step
    if (ctx->flow_err && *(volatile int32_t *)ctx->flow_err != 0) { // (the flag lives in page-locked memory the kernel writes itself)
        *(volatile int32_t *)ctx->flow_err = 0;
        ctx->flow_hist_at = nullptr; // whatever the histories hold now: sentinel again before the next use
        ctx->lazy_wpart_dirty = true;  // (and lazy_window's polled partials)
        return crec_fail(ctx, CREC_EHIP, "an earlier bpr_steps_flow (or lazy_window) launch gave up waiting for a row version (a wave of it never ran or the "
                                         "device faulted): the tables that call left are not valid");
    }
    const auto ensure_err = [&]() { // the give-up flag of the launches whose waves wait for each other: page-locked, mapped
        if (ctx->flow_err)
            return;
        void *dev = nullptr;
        if (hipHostMalloc((void **)&ctx->flow_err, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
            hipHostGetDevicePointer(&dev, ctx->flow_err, 0) == hipSuccess) {
            *ctx->flow_err = 0;
            ctx->flow_err_dev = (int32_t *)dev;
        } else {
            (void)hipGetLastError();
            if (ctx->flow_err)
                (void)hipHostFree(ctx->flow_err);
            ctx->flow_err = nullptr;
        }
    };
    FlowKernel fk{};
    int flow_cmax = 0; // steps per launch; 0: bpr_step_fused, a launch per step
    int CH_idx = CH;   // what the index kernel cuts heavy rows at
    const int64_t hP = (int64_t)t->U * t->d, hQ = (int64_t)t->I * t->d;
    if (dense && !dp && ctx->knob_flow != 0) {
        fk = pick_flow(t->d);
        const int64_t grid = crec_ceil_div(t->U, fk.rpb_u) + crec_ceil_div(t->I, fk.rpb_q);
        int &bpc = ctx->flow_blocks_per_cu[fk.lpr_log2][fk.full];
        if (bpc == 0) {
            int nbk = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbk, fk.fn, CREC_FUSED_BLOCK, 0) != hipSuccess) {
                (void)hipGetLastError();
                nbk = 0;
            }
            bpc = nbk > 0 ? nbk : -1;
        }
        const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
        ensure_err();
        if (bpc > 0 && grid <= (int64_t)bpc * cus && ctx->flow_err_dev && hP * 4 < ((int64_t)1 << 31) && hQ * 4 < ((int64_t)1 << 31)) {
            int64_t cmax = FLOW_MAX_STEPS;
            while (cmax > 1 && 2 * (cmax - 1) * (hP + hQ) * 4 > ((int64_t)1 << 30))
                cmax /= 2;
            if (ctx->knob_flow_steps > 0)
                cmax = ctx->knob_flow_steps < cmax ? ctx->knob_flow_steps : cmax;
            if (cmax >= 8 || ctx->knob_flow_steps > 0)
                flow_cmax = (int)cmax;
        }
    }
    const int flow_rpb = flow_cmax ? (fk.rpb_u | (fk.rpb_q << 8)) : 0;
    if (flow_cmax) { // its waves walk whatever their rows hold: no heavy-row chunks, no wide groups, nothing for the host to read back
        CH_idx = 0x3fffffff;
        wide_total = 0;
    }
    if (flow_cmax > 1) {
        const int64_t slots = flow_cmax - 1;
        const size_t bytes = sizeof(float) * (size_t)(2 * slots * (hP + hQ));
        if (int rc = crec_reserve(ctx, &ctx->flow_hist, bytes))
            return rc;
        if (ctx->flow_hist_at != ctx->flow_hist.ptr || ctx->flow_key[0] != flow_cmax || ctx->flow_key[1] != hP || ctx->flow_key[2] != hQ) {
            CREC_HIP(ctx, hipMemsetAsync(ctx->flow_hist.ptr, 0xff, bytes, (hipStream_t)stream)); // the sentinel in every word
            ctx->flow_hist_at = ctx->flow_hist.ptr;
            ctx->flow_key[0] = flow_cmax, ctx->flow_key[1] = hP, ctx->flow_key[2] = hQ;
            ctx->flow_dirty[0] = ctx->flow_dirty[1] = 0;
            ctx->flow_cur = 0;
        }
    }
    // the row dealing of bpr_steps_flow by weight: made from this call's triples when the ctx has none for the shape (once: a count
    // kernel, one read-back, the host's dealing, one upload)
    const int32_t *deal[4] = {nullptr, nullptr, nullptr, nullptr}; // pos_u, row_u, pos_q, row_q
    int deal_nwg_u = 0, deal_nwg_q = 0;
    if (flow_cmax && ctx->knob_flow_deal && t->U <= FLOW_DEAL_MAX && t->I <= FLOW_DEAL_MAX) {
        const auto more = [](int64_t n_, int rpb_) { const int64_t base_ = crec_ceil_div(n_, rpb_); return (int)(base_ + base_ / 12 + 1); };
        const int nwg_u = more(t->U, fk.rpb_u), nwg_q = more(t->I, fk.rpb_q);
        const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
        if ((int64_t)nwg_u + nwg_q <= (int64_t)ctx->flow_blocks_per_cu[fk.lpr_log2][fk.full] * cus) { // (else: the plain dealing, which fits)
            const size_t n_cnt = align_up((size_t)(t->U + t->I), 64), n_pu = align_up((size_t)t->U, 64), n_ru = align_up((size_t)nwg_u * fk.rpb_u, 64),
                         n_pq = align_up((size_t)t->I, 64), n_rq = align_up((size_t)nwg_q * fk.rpb_q, 64);
            const void *before = ctx->flow_deal.ptr;
            if (int rc = crec_reserve(ctx, &ctx->flow_deal, sizeof(int32_t) * (n_cnt + n_pu + n_ru + n_pq + n_rq)))
                return rc;
            int32_t *cnt = (int32_t *)ctx->flow_deal.ptr, *pos_u = cnt + n_cnt, *row_u = pos_u + n_pu, *pos_q = row_u + n_ru, *row_q = pos_q + n_pq;
            if (ctx->flow_deal.ptr != before || ctx->flow_deal_key[0] != t->U || ctx->flow_deal_key[1] != t->I || ctx->flow_deal_key[2] != flow_rpb) {
                hipStream_t st_ = (hipStream_t)stream;
                CREC_HIP(ctx, hipMemsetAsync(cnt, 0, sizeof(int32_t) * n_cnt, st_));
                const int64_t n_w = N < (int64_t)256 * B ? N : (int64_t)256 * B; // (a few hundred batches say all there is to know)
                hipLaunchKernelGGL(flow_count_rows, dim3((unsigned)(crec_ceil_div(n_w, 256) < 1024 ? crec_ceil_div(n_w, 256) : 1024)), dim3(256), 0, st_, u,
                                   i, j, n_w, cnt, cnt + t->U);
                std::vector<int32_t> h_cnt((size_t)(t->U + t->I)), h_tab(n_pu + n_ru + n_pq + n_rq);
                CREC_HIP(ctx, hipMemcpyAsync(h_cnt.data(), cnt, sizeof(int32_t) * h_cnt.size(), hipMemcpyDeviceToHost, st_));
                CREC_HIP(ctx, hipStreamSynchronize(st_));
                flow_deal_host(h_cnt.data(), (int)t->U, nwg_u, fk.rpb_u, h_tab.data(), h_tab.data() + n_pu);
                flow_deal_host(h_cnt.data() + t->U, (int)t->I, nwg_q, fk.rpb_q, h_tab.data() + n_pu + n_ru, h_tab.data() + n_pu + n_ru + n_pq);
                CREC_HIP(ctx, hipMemcpyAsync(pos_u, h_tab.data(), sizeof(int32_t) * h_tab.size(), hipMemcpyHostToDevice, st_));
                CREC_HIP(ctx, hipStreamSynchronize(st_)); // (h_tab is pageable and goes out of scope)
                ctx->flow_deal_key[0] = t->U, ctx->flow_deal_key[1] = t->I, ctx->flow_deal_key[2] = flow_rpb;
            }
            deal[0] = pos_u, deal[1] = row_u, deal[2] = pos_q, deal[3] = row_q;
            deal_nwg_u = nwg_u, deal_nwg_q = nwg_q;
        }
    }
    // lazy-family steps, W consecutive batches per launch (bpr_triples_lazy<.., WIN>): where the batch is indexed in LDS and the step is
    // not the row-sharded one (deferred Adam's per-row step stamps travel like the rows: read where the XCDs meet, written through)
    int lazy_win = 0;
    if (!dense && !dp && B <= MAX_B && !ctx->knob_big_index && ctx->knob_lazy_window != 0 && ctx->knob_lazy_window != 1 &&
        3 * (int64_t)B < ((int64_t)1 << 20)) {
        // -1 (the default): 12 batches per launch where the tables outgrow the batch -- a row then seldom has a predecessor in the
        // launch and the launch is bound by its traffic, not by the popular rows' way from batch to batch.  us per step, a launch per
        // batch / 8 / 12 per launch (tools/time_window.py): 10 M x 1 M rows, d = 128: 18.6 / 13.9 / 13.4 (uniform positives 21.4 /
        // 14.8 / 13.7; B = 2048: 12.1 / 8.4 / 8.1), d = 32: 12.2 / 9.6 / 9.8; 1 M x 100 k: 17.9 / 14.6 / 14.3 (B = 1024: 10.6 /
        // 7.2 / 6.8); 100 k x 20 k, d = 64: 13.7 / 13.0 / 13.0; but ml-1m's 6 040 x 3 706, where every row recurs in every batch:
        // d = 64 13.7 / 15.7 / 15.7 (d = 128: 17.6 / 17.0 / 16.9)
        int want = ctx->knob_lazy_window;
        if (want < 0)
        {
            // (deferred Adam: a row's replay sits on its way from batch to batch as well -- from 16 batches' worth of rows only:
            // 50 k x 20 k rows 23.2 -> 26.4 us per step under the window, 500 k x 100 k 47.3 -> 45.6)
            const int64_t need = (df ? 16 : 3) * (int64_t)B;
            want = (t->d >= 32 && (int64_t)t->I >= need && (int64_t)t->U >= need && nb >= 4) ? 12 : 0;
        }
        if (want > 1) {
            ensure_err();
            if (ctx->flow_err_dev)
                lazy_win = want < LAZY_WIN_MAX ? want : LAZY_WIN_MAX;
        }
    }
    const auto layout = [&](void *base, int64_t nbc_, int spw_ = 32) {
        IndexWs w_ = layout_index(base, nbc_, B, t->U, t->I, dense, CH_idx, wide_rows, wide_total, spw_, flow_rpb, deal_nwg_u, deal_nwg_q, lazy_win);
        w_.flow_pos_u = deal[0], w_.flow_row_u = deal[1], w_.flow_pos_q = deal[2], w_.flow_row_q = deal[3];
        return w_;
    };

    // chunk the epoch so the inverted indices of a chunk stay under ~1 GiB
    const size_t per_batch = layout(nullptr, 1).bytes;
    int64_t chunk = (int64_t)((size_t)1 << 30) / (int64_t)per_batch;
    // batches beyond the LDS sort (and any batch under the "big_index" option) are indexed in global memory, a chunk at a time
    const bool big_index = !dense && (B > MAX_B || ctx->knob_big_index);
    if (big_index && chunk * B > ((int64_t)1 << 22)) // its scratch: ~100 bytes per triple of the chunk
        chunk = ((int64_t)1 << 22) / B;
    if (big_index && chunk > 4096) // ... and 2 KB of digit counters per batch and sort tile
        chunk = 4096;
    chunk = chunk < 1 ? 1 : (chunk > nb ? nb : chunk);
    // The chunks' sizes.  The global-memory index does not depend on the tables, so under the "idx_overlap" option chunk c + 1 is
    // indexed on a side stream while chunk c's steps run (two index buffers); the first chunk's index is then the only one the
    // steps wait for, so the chunks start small (half a million triples) and double up to `chunk`.  (Off by default: S-big at
    // B = 65536 on one GPU, 135.1 us per step so against 132.8 in line -- the step kernel is slowed by what the index kernels take.)
    // The lazy family's LDS-built index (and lazy_window's dependencies) can go the same way -- nothing of it is read back by the host
    // and its steps wait for nobody dispatched later, so an index kernel beside them is safe (not so beside bpr_steps_flow, whose
    // waves wait for each other and must all be resident) -- and gains as little: S-big 15.7 us per step so against 15.65 in line,
    // the window launch 14.3 us per batch instead of 13.0 (the index's workgroups take the CUs the steps would have used).
    const bool want_pipe = ctx->knob_idx_overlap == 1 && (big_index || (!dense && !dp));
    std::vector<int64_t> chunk_sizes;
    {
        int64_t cur = chunk;
        if (want_pipe) {
            cur = ((int64_t)1 << 19) / B;
            cur = cur < 1 ? 1 : (cur > chunk ? chunk : cur);
        }
        for (int64_t done = 0; done < nb; done += cur, cur = (2 * cur > chunk ? chunk : 2 * cur))
            chunk_sizes.push_back((nb - done) < cur ? (nb - done) : cur);
    }
    const bool piped = want_pipe && chunk_sizes.size() > 1;
    if (int rc = crec_reserve(ctx, &ctx->idx_ws, layout(nullptr, chunk).bytes))
        return rc;
    if (piped) {
        if (int rc = crec_reserve(ctx, &ctx->idx_ws2, layout(nullptr, chunk).bytes))
            return rc;
        hipStream_t side_;
        if (int rc = crec_side_stream(ctx, &side_))
            return rc;
    }
    // per batch: x_b of every triple, then one l2 term per row wave
    const int64_t loss_stride = B + (dense ? t->U + t->I : (int64_t)B); // lazy: x_b, then the l2 term of every triple
    if (int rc = crec_reserve(ctx, &ctx->loss_ws, sizeof(float) * (size_t)chunk * loss_stride))
        return rc;
    const IndexWs w0 = layout(nullptr, 1);
    // heavy rows: chunk partials, then one ticket counter per heavy row (zeroed here, re-zeroed by the finishing wave)
    const size_t part_bytes = align_up(sizeof(float) * (size_t)(w0.HIu + w0.HIq) * t->d, 256);
    const size_t ticket_bytes = sizeof(int32_t) * (size_t)(w0.HRu + w0.HRq);
    if (int rc = crec_reserve(ctx, &ctx->misc, part_bytes + ticket_bytes))
        return rc;
    // the finishing wave of a heavy row re-zeroes its counter, so the counters only need clearing when they move
    if (ticket_bytes && (ctx->tickets_at != (char *)ctx->misc.ptr + part_bytes || ctx->tickets_bytes != ticket_bytes)) {
        CREC_HIP(ctx, hipMemsetAsync((char *)ctx->misc.ptr + part_bytes, 0, ticket_bytes, st));
        ctx->tickets_at = (char *)ctx->misc.ptr + part_bytes;
        ctx->tickets_bytes = ticket_bytes;
    }

    StepArgs a{};
    a.ctx = ctx;
    a.wide_round = ctx->knob_wide_round;
    a.cur = Tables{t->P, t->mP, t->vP, t->Q, t->mQ, t->vQ};
    a.d = t->d;
    a.U = t->U;
    a.I = t->I;
    a.B = B;
    a.st = st;
    a.partial = (float *)ctx->misc.ptr;
    a.tickets = (int32_t *)((char *)ctx->misc.ptr + part_bytes);
    const size_t bytesP = sizeof(float) * (size_t)t->U * t->d, bytesQ = sizeof(float) * (size_t)t->I * t->d;
    if (dense) { // the second buffer set of the ping-pong
        const size_t aP = align_up(bytesP, 256), aQ = align_up(bytesQ, 256);
        if (int rc = crec_reserve(ctx, &ctx->shadow, 3 * aP + 3 * aQ))
            return rc;
        char *s = (char *)ctx->shadow.ptr;
        a.other = Tables{(float *)s, (float *)(s + aP), (float *)(s + 2 * aP), (float *)(s + 3 * aP),
                         (float *)(s + 3 * aP + aQ), (float *)(s + 3 * aP + 2 * aQ)};
        if (dp) { // Q does not ping-pong: the fused launch only emits dQ, bpr_adam_dense updates Q in place
            a.other.Q = a.cur.Q;
            a.other.mQ = a.cur.mQ;
            a.other.vQ = a.cur.vQ;
            if (int rc = crec_reserve(ctx, &ctx->dq, bytesQ))
                return rc;
        }
    } else {
        if (int rc = setup_lazy_bufs(ctx, B, t->d, CH, st, &a.lazy, lazy_win ? lazy_win : 1))
            return rc;
        if (lazy_win) { // the stamps by (batch in window, segment start): zero once -- a stamp is a ctx-wide launch count and never repeats
            const size_t bytes = sizeof(int32_t) * (size_t)LAZY_WIN_MAX * 3 * (size_t)B;
            const void *before = ctx->lazy_done.ptr;
            if (int rc = crec_reserve(ctx, &ctx->lazy_done, bytes))
                return rc;
            if (ctx->lazy_done.ptr != before)
                CREC_HIP(ctx, hipMemsetAsync(ctx->lazy_done.ptr, 0, ctx->lazy_done.bytes, st));
        }
        a.lazy.wpart = nullptr, a.lazy.err = nullptr;
        if (lazy_win && !(ctx->knob_flow_abl & 8)) {
            // the polled partials of front rows (lazy_poll_front): all bits set = nothing waits; whoever takes a partial puts that back.
            // (Window launches only.  A launch per batch loses by it -- ml-1m's catalogue 13.7 -> 14.8 us per step, S-big 18.6 -> 19.1:
            // all its waves start together, and a finisher that waits holds its place while the body is still under way.)
            ensure_err();
            const size_t pbytes = sizeof(float) * (size_t)(lazy_win ? lazy_win : 1) * 2 * (size_t)B * (size_t)t->d;
            const void *pbefore = ctx->lazy_wpart.ptr;
            const size_t phad = ctx->lazy_wpart.bytes;
            if (int rc = crec_reserve(ctx, &ctx->lazy_wpart, pbytes))
                return rc;
            if (ctx->lazy_wpart.ptr != pbefore || ctx->lazy_wpart.bytes != phad || ctx->lazy_wpart_dirty) {
                CREC_HIP(ctx, hipMemsetAsync(ctx->lazy_wpart.ptr, 0xff, ctx->lazy_wpart.bytes, st));
                ctx->lazy_wpart_dirty = false;
            }
            if (ctx->flow_err_dev)
                a.lazy.wpart = (float *)ctx->lazy_wpart.ptr, a.lazy.err = ctx->flow_err_dev;
        }
    }

    bool in_shadow = false; // tf1: which buffer set holds the current tables
    // Whatever way this function is left -- also through one of the error returns below -- the caller's buffers must hold
    // the newest tables: after an odd number of completed steps they live in the ctx's second buffer set
    struct BringHome {
        const crec_bpr_tables *t;
        const StepArgs *a;
        const bool *in_shadow;
        size_t bytesP, bytesQ;
        hipStream_t st;
        // the copies that bring the current tables home (0 when they are at home): the tables' description and how many
        int describe(CopyTables *c, int64_t *most) const
        {
            if (!*in_shadow)
                return 0;
            const Tables &s = a->cur;
            const bool q_too = s.Q != t->Q;
            float *const dst[6] = {t->P, t->mP, t->vP, t->Q, t->mQ, t->vQ};
            const float *const src[6] = {s.P, s.mP, s.vP, s.Q, s.mQ, s.vQ};
            *c = CopyTables{};
            *most = 0;
            for (int k = 0; k < (q_too ? 6 : 3); ++k) {
                c->dst[k] = reinterpret_cast<float4 *>(dst[k]), c->src[k] = reinterpret_cast<const float4 *>(src[k]);
                c->n4[k] = (int64_t)((k < 3 ? bytesP : bytesQ) / sizeof(float4));
                *most = c->n4[k] > *most ? c->n4[k] : *most;
            }
            return q_too ? 6 : 3;
        }
        int run() const
        {
            // ONE launch for the six tables: six device copies of 1.5 / 0.9 MB each took 4.9 us apiece behind a 20-batch call
            // (29 of its 177 us)
            CopyTables c{};
            int64_t most = 0;
            const int n = describe(&c, &most);
            if (n == 0)
                return (int)hipSuccess;
            const unsigned bx = (unsigned)(crec_ceil_div(most, 256) < 256 ? crec_ceil_div(most, 256) : 256);
            hipLaunchKernelGGL(copy_tables, dim3(bx ? bx : 1, (unsigned)n), dim3(256), 0, st, c);
            return (int)hipGetLastError();
        }
        bool done = false;
        ~BringHome()
        {
            if (!done)
                (void)run(); // an error return: best effort, the error code of the failing call is what the caller sees
        }
    } home{t, &a, &in_shadow, bytesP, bytesQ, st};
    const auto build_big = [&](const IndexWs &wb, int64_t first, int64_t count, hipStream_t s) {
        const int64_t tb = first * B, nc = (N - tb) < count * B ? (N - tb) : count * B;
        const LazyIndexOut o{wb.u_desc, wb.i_desc, wb.j_desc, wb.i_tmp, wb.j_tmp, wb.slot_pu, wb.order, wb.nslots};
        return crec_lazy_index_big(ctx, u + tb, i + tb, j + tb, nc, B, count, wb.bits_u, wb.bits_q, wb.spw, o, s, chunk);
    };
    // dynamic LDS of the index kernel
    const int cap = (2 * B + 63) / 64 * 64;
    // (tf1: the ids in LDS where the catalogues fit 16 bits and the workgroup's LDS the three streams: B <= 6 144)
    const bool ids16 = dense && !big_index && t->U < 65536 && t->I < 65536 && index_lds_bytes(cap, B) <= (size_t)160 * 1024 && !ctx->knob_idx_no_ids16;
    const size_t lds = big_index ? 0 : index_lds_bytes(cap, ids16 ? B : 0);
    if (lds > ctx->index_lds_set) { // a driver call per epoch otherwise
        CREC_HIP(ctx, hipFuncSetAttribute((const void *)bpr_index_batches, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)lds));
        ctx->index_lds_set = lds;
    }
    const auto build_lds = [&](const IndexWs &wb, int64_t first, int64_t count, hipStream_t s) -> int {
        const int64_t tb = first * B, nc = (N - tb) < count * B ? (N - tb) : count * B;
        if (lazy_win) // (the presence filters are OR-ed into)
            CREC_HIP(ctx, hipMemsetAsync(wb.wfilt, 0, sizeof(uint32_t) * (size_t)count * 2 * WF_WORDS, s));
        crec_prof_begin(ctx, CREC_K_INDEX, s);
        // tf1, a call of few batches: two workgroups per (batch, side) share the phases behind the sort -- as long as all of them
        // find a CU of their own at once it is the latency of one workgroup that the steps wait for (a 20-batch call: 524 ->
        // 536-553 M triples/s); an epoch's 455 batches fill the chip anyway, and the second sort is then just more work (810 -> 785 M)
        const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
        const int idx_parts = dense && !ctx->knob_idx_one_wg && 4 * count <= cus ? 2 : 1;
        IndexWs wk = wb;
        wk.ids16 = ids16 ? 1 : 0;
        CREC_KLAUNCH(ctx, bpr_index_batches, dim3((unsigned)(2 * idx_parts * count)), dim3(PREP_THREADS), lds, s, u + tb, i + tb,
                           j + tb, nc, B, cap, wk, idx_parts);
        crec_prof_end(ctx, s);
        if (lazy_win)
            hipLaunchKernelGGL(lazy_window_deps, dim3((unsigned)crec_ceil_div(B, 256), (unsigned)count), dim3(256), 0, s, wb, lazy_win);
        return CREC_OK;
    };
    const auto build_any = [&](const IndexWs &wb, int64_t first, int64_t count, hipStream_t s) -> int {
        return big_index ? build_big(wb, first, count, s) : build_lds(wb, first, count, s);
    };
    size_t ci = 0;
    for (int64_t b0 = 0, nbc = 0; b0 < nb; b0 += nbc, ++ci) {
        const crec_range chunk_range("crec_bpr chunk of batches");
        nbc = chunk_sizes[ci];
        const int64_t t0 = b0 * B, n_chunk = (N - t0) < nbc * B ? (N - t0) : nbc * B;
        void *const ws_of[2] = {ctx->idx_ws.ptr, piped ? ctx->idx_ws2.ptr : ctx->idx_ws.ptr};
        IndexWs w = layout(ws_of[ci & 1], nbc, lazy_spw(t->d, B));
        const size_t nw_off = (size_t)(w.nwide - w.hn), n_back = dense ? nw_off + (size_t)nbc * 2 : 0;
        if (n_back + 64 > ctx->pinned_ints) {
            if (ctx->pinned)
                CREC_HIP(ctx, hipHostFree(ctx->pinned));
            ctx->pinned = nullptr;
            ctx->pinned_ints = 0;
            ctx->pinned_dev = nullptr;
            // mapped + coherent: the index kernel's last workgroup writes the counts and the flag here itself
            const size_t want = sizeof(int32_t) * (n_back * 2 + 64);
            bool mapped = hipHostMalloc((void **)&ctx->pinned, want, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess;
            if (!mapped) { // no coherent mapping on this system: plain page-locked memory, the counts come back by copy
                (void)hipGetLastError();
                ctx->pinned = nullptr;
                CREC_HIP(ctx, hipHostMalloc((void **)&ctx->pinned, want, hipHostMallocDefault));
            }
            ctx->pinned_ints = n_back * 2 + 64;
            memset(ctx->pinned, 0, sizeof(int32_t) * ctx->pinned_ints);
            void *dev = nullptr;
            if (mapped && hipHostGetDevicePointer(&dev, ctx->pinned, 0) == hipSuccess)
                ctx->pinned_dev = (int32_t *)dev;
            else
                (void)hipGetLastError();
        }
        if (!ctx->idx_done) {
            CREC_HIP(ctx, hipMalloc((void **)&ctx->idx_done, sizeof(int32_t)));
            CREC_HIP(ctx, hipMemsetAsync(ctx->idx_done, 0, sizeof(int32_t), st));
        }
        // the flag sits in the buffer's last cache line, the counts at its start
        const size_t flag_at = ctx->pinned_ints - 16;
        const bool zero_copy = dense && !flow_cmax && ctx->pinned_dev && !ctx->knob_no_zero_copy;
        w.host_counts = zero_copy ? ctx->pinned_dev : nullptr;
        w.host_flag = zero_copy ? ctx->pinned_dev + flag_at : nullptr;
        w.done = ctx->idx_done;
        w.seq = zero_copy ? ++ctx->idx_seq : 0;
        w.n_back = (int32_t)n_back;
        if (!piped) {
            if (int rc = build_any(w, b0, nbc, st))
                return rc;
        } else {
            hipStream_t side = ctx->idx_stream;
            if (ci == 0) { // the side stream starts behind whatever the caller's stream holds (the triples' producer, an earlier epoch's steps)
                CREC_HIP(ctx, hipEventRecord(ctx->idx_ev[4], st));
                CREC_HIP(ctx, hipStreamWaitEvent(side, ctx->idx_ev[4], 0));
                if (int rc = build_any(w, b0, nbc, side))
                    return rc;
                CREC_HIP(ctx, hipEventRecord(ctx->idx_ev[0], side));
            }
            if (ci + 1 < chunk_sizes.size()) { // the next chunk's index, beside this chunk's steps, into the other buffer
                IndexWs wn = layout(ws_of[(ci + 1) & 1], chunk_sizes[ci + 1], lazy_spw(t->d, B));
                wn.done = ctx->idx_done; // (no counts go back to the host on these paths)
                if (ci >= 1) // (that buffer was read by the steps of chunk ci - 1)
                    CREC_HIP(ctx, hipStreamWaitEvent(side, ctx->idx_ev[2 + ((ci + 1) & 1)], 0));
                if (int rc = build_any(wn, b0 + nbc, chunk_sizes[ci + 1], side))
                    return rc;
                CREC_HIP(ctx, hipEventRecord(ctx->idx_ev[(ci + 1) & 1], side));
            }
            CREC_HIP(ctx, hipStreamWaitEvent(st, ctx->idx_ev[ci & 1], 0));
        }
        // tf1 mode: the chunk's counts of heavy-row chunks and wide groups size the launches.  They arrive in page-locked
        // memory written by the kernel itself (above); the host polls the flag -- a device-to-host copy plus a stream
        // synchronisation cost ~15 us per call, a tenth of a 20-batch call.  (Lazy mode has nothing to read back: its
        // launches are sized by the batch alone, so the whole epoch is enqueued without a host synchronisation.)
        const int32_t *hn = ctx->pinned, *nwide = ctx->pinned + nw_off;
        a.nwide = nwide;
        if (dense && !flow_cmax) {
            bool have = false;
            if (zero_copy) {
                volatile int32_t *flag = ctx->pinned + flag_at;
                const auto t_give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(200);
                unsigned spins = 0;
                while (!(have = (*flag == w.seq))) {
                    __builtin_ia32_pause();
                    if ((++spins & 1023) == 0 && std::chrono::steady_clock::now() > t_give_up)
                        break; // (a profiler serialising the launch, a device error: fall back to the copy)
                }
                std::atomic_thread_fence(std::memory_order_acquire);
                for (int64_t b = 0; have && b < nbc; ++b) // every workgroup's stamp: its counts are this launch's
                    have = ctx->pinned[(size_t)b * HN + 6] == w.seq && ctx->pinned[(size_t)b * HN + 7] == w.seq;
            }
            if (!have) {
                CREC_HIP(ctx, hipMemcpyAsync(ctx->pinned, w.hn, sizeof(int32_t) * n_back, hipMemcpyDeviceToHost, st));
                CREC_HIP(ctx, hipStreamSynchronize(st));
            } else {
                // the chunk's first step is enqueued while the index kernel is still retiring: the two events attached
                // to that dispatch then also span the wait (measured: 8.5-10 us "per launch" over a 20-step call against
                // 6.4) -- the per-kernel profile leaves that one launch out
                ctx->prof.skip_next = ctx->prof.on;
            }
        }
        a.pad_wide = a.pad_chunks_u = a.pad_chunks_q = 0;
        for (int64_t b = 0; b < (dense && !flow_cmax ? nbc : 0); ++b) {
            const int nw = nwide[(size_t)b * 2] + nwide[(size_t)b * 2 + 1];
            a.pad_wide = nw > a.pad_wide ? nw : a.pad_wide;
            a.pad_chunks_u = hn[(size_t)b * HN + 1] > a.pad_chunks_u ? hn[(size_t)b * HN + 1] : a.pad_chunks_u;
            a.pad_chunks_q = hn[(size_t)b * HN + 3] > a.pad_chunks_q ? hn[(size_t)b * HN + 3] : a.pad_chunks_q;
        }
        for (int64_t b = 0; flow_cmax && b < nbc;) { // bpr_steps_flow: up to flow_cmax steps per launch
            const int C = (int)((nbc - b) < flow_cmax ? (nbc - b) : flow_cmax);
            FlowArgs f{};
            f.u_rowseg = w.u_rowseg + b * w.U;
            f.q_rowseg = w.q_rowseg + b * w.I;
            f.u_occ = w.u_occ + b * w.B;
            f.q_occ = w.q_occ + b * 2 * w.B;
            f.Upos = w.U, f.Ipos = w.I, f.B = B;
            f.nwg_u = w.flow_nwg_u, f.nwg_q = w.flow_nwg_q;
            f.row_u = w.flow_row_u, f.row_q = w.flow_row_q;
            f.C = C;
            f.hP = hP, f.hQ = hQ;
            const int64_t slots = ctx->flow_hist_at ? ctx->flow_key[0] - 1 : 0;
            const int cur = ctx->flow_cur;
            if (slots > 0) {
                float *base = (float *)ctx->flow_hist.ptr;
                f.histP = base + (int64_t)cur * slots * (hP + hQ);
                f.histQ = f.histP + slots * hP;
                f.nextP = base + (int64_t)(1 - cur) * slots * (hP + hQ);
                f.nextQ = f.nextP + slots * hP;
                f.dirty_next = ctx->flow_dirty[1 - cur];
            }
            f.loss_ws = (float *)ctx->loss_ws.ptr + b * loss_stride;
            f.loss_stride = loss_stride;
            f.err = ctx->flow_err_dev;
            f.abl = ctx->knob_flow_abl;
            for (int k = 0; k < C; ++k) { // lr_t of every step of the launch (fp32 running powers, as below)
                f.lr_t[k] = (opt->lr * sqrtf(1.0f - opt->beta2_power)) / (1.0f - opt->beta1_power);
                opt->beta1_power = opt->beta1_power * opt->beta1;
                opt->beta2_power = opt->beta2_power * opt->beta2;
                opt->step += 1;
            }
            a.h = Hyper{};
            a.h.lr_t = f.lr_t[0];
            a.h.beta1 = opt->beta1;
            a.h.beta2 = opt->beta2;
            a.h.eps = opt->eps;
            a.h.reg = opt->reg;
            ctx->prof.next_units = C;
            crec_prof_begin(ctx, CREC_K_STEPS_FLOW, st);
            fk.launch(ctx, (unsigned)(w.flow_nwg_u + w.flow_nwg_q), st, a.cur, a.other, a.d, a.U, a.I, f, a.h);
            crec_prof_end(ctx, st);
            ctx->prof.next_units = 1;
            if (slots > 0) {
                ctx->flow_dirty[cur] = C - 1;  // versions 1 .. C - 1 of this launch
                ctx->flow_dirty[1 - cur] = 0;  // (put back to the sentinel by this launch)
                ctx->flow_cur = 1 - cur;
            }
            Tables tmp = a.cur;
            a.cur = a.other;
            a.other = tmp;
            in_shadow = !in_shadow;
            b += C;
        }
        for (int64_t b = 0; lazy_win && b < nbc;) { // W batches per launch (windows are aligned to the chunk's first batch, as lazy_window_deps has them)
            const int Wc = (int)((nbc - b) < lazy_win ? (nbc - b) : lazy_win);
            LazyWin wn{};
            wn.W = Wc;
            wn.B = B;
            wn.n_left = n_chunk - b * B;
            wn.loss_stride = loss_stride;
            wn.z_stride = 6 * (int64_t)B;
            wn.deps = w.wdeps + b * B;
            wn.done = (int32_t *)ctx->lazy_done.ptr;
            wn.stamp0 = ctx->lazy_stamp + 1;
            ctx->lazy_stamp += LAZY_WIN_MAX;
            if (ctx->lazy_stamp > 0x7fff0000)
                ctx->lazy_stamp = 0; // (old stamps are at least a launch behind: never equal to a fresh one)
            wn.err = ctx->flow_err_dev;
            wn.abl = ctx->knob_flow_abl;
            wn.fit_p = hP * 4 <= (int64_t)0xfffffff0, wn.fit_q = hQ * 4 <= (int64_t)0xfffffff0;
            a.h = Hyper{};
            a.h.opt = (opt->adam_mode == CREC_OPT_SGD || opt->adam_mode == CREC_OPT_ADAGRAD) ? opt->adam_mode : 0;
            a.h.beta1 = opt->beta1, a.h.beta2 = opt->beta2, a.h.eps = opt->eps, a.h.reg = opt->reg;
            a.h.lr_tab = lr_tab; // deferred mode (else null)
            a.h.lastP = df ? df->lastP : nullptr;
            a.h.lastQ = df ? df->lastQ : nullptr;
            wn.step0 = (int32_t)(opt->step + 1);
            a.h.step = wn.step0;
            for (int k = 0; k < Wc; ++k) {
                wn.lr_t[k] = a.h.opt ? opt->lr : (opt->lr * sqrtf(1.0f - opt->beta2_power)) / (1.0f - opt->beta1_power);
                opt->beta1_power = opt->beta1_power * opt->beta1;
                opt->beta2_power = opt->beta2_power * opt->beta2;
                opt->step += 1;
            }
            a.h.lr_t = wn.lr_t[0];
            a.xb = (float *)ctx->loss_ws.ptr + b * loss_stride;
            a.rowterm = a.xb + B;
            pick_window(t->d)(a, w, b, wn);
            b += Wc;
        }
        for (int64_t b = 0; !flow_cmax && !lazy_win && b < nbc; ++b) {
            // lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t) with fp32 running powers (TF1 _prepare/_finish); SGD / Adagrad: lr
            a.h.opt = (opt->adam_mode == CREC_OPT_SGD || opt->adam_mode == CREC_OPT_ADAGRAD) ? opt->adam_mode : 0;
            a.h.lr_tab = lr_tab; // deferred mode (else null)
            a.h.lastP = df ? df->lastP : nullptr;
            a.h.lastQ = df ? df->lastQ : nullptr;
            a.h.step = (int32_t)(opt->step + 1);
            a.h.lr_t = a.h.opt ? opt->lr : (opt->lr * sqrtf(1.0f - opt->beta2_power)) / (1.0f - opt->beta1_power);
            a.h.beta1 = opt->beta1;
            a.h.beta2 = opt->beta2;
            a.h.eps = opt->eps;
            a.h.reg = opt->reg;
            // (the peer transport: dQ goes straight into this rank's window, and ONE launch exchanges it and updates Q)
            const bool dp_fused = dp && !ctx->knob_dp_unfused && xdp->has_allreduce_adam((size_t)t->I * t->d);
            a.h.dQ = dp ? (dp_fused ? (float *)xdp->next_half() : (float *)ctx->dq.ptr) : nullptr;
            a.h.dq_sys = dp_fused;
            a.xb = (float *)ctx->loss_ws.ptr + b * loss_stride;
            a.nb_cur = (int32_t)((n_chunk - b * B) < B ? (n_chunk - b * B) : B);
            a.rowterm = a.xb + B;
            step(a, w, b, hn + b * HN);
            if (dp && dp_fused) {
                const crec_adam_args aa{a.h.lr_t, a.h.beta1, a.h.beta2, a.h.eps};
                crec_prof_begin(ctx, CREC_K_PEER_ADAM, st);
                const int rc = xdp->allreduce_adam(ctx, (size_t)t->I * t->d, t->Q, t->mQ, t->vQ, aa, st);
                crec_prof_end(ctx, st);
                if (rc)
                    return rc;
            } else if (dp) {
                // the one exchange of the path: per-item gradient sums over all ranks' triples
                float *dq = (float *)ctx->dq.ptr;
                const size_t nq = (size_t)t->I * t->d;
                if (int rc = xdp->allreduce_sum_f32(ctx, dq, nq, st))
                    return rc;
                crec_prof_begin(ctx, CREC_K_ADAM_DENSE, st);
                const int64_t n4 = (int64_t)nq / 4;
                const unsigned blocks = (unsigned)(crec_ceil_div(n4, 256) < 2048 ? crec_ceil_div(n4, 256) : 2048);
                CREC_KLAUNCH(ctx, bpr_adam_dense, dim3(blocks), dim3(256), 0, st, t->Q, t->mQ, t->vQ, (const float *)dq, n4,
                                   a.h);
                crec_prof_end(ctx, st);
            }
            if (dense) {
                Tables tmp = a.cur;
                a.cur = a.other;
                a.other = tmp;
                in_shadow = !in_shadow;
            }
            opt->beta1_power = opt->beta1_power * opt->beta1;
            opt->beta2_power = opt->beta2_power * opt->beta2;
            opt->step += 1;
        }
        double *l_parts;
        int32_t *l_tickets;
        if (int rc = loss_parts(ctx, chunk, st, &l_parts, &l_tickets))
            return rc;
        // the call's last chunk, its tables in the second buffer set: their way home rides in the reduction's launch
        CopyTables home_c{};
        int64_t home_most = 0;
        const int home_n = b0 + nbc >= nb && !ctx->knob_no_call_tail ? home.describe(&home_c, &home_most) : 0;
        crec_prof_begin(ctx, CREC_K_REDUCE_LOSS, st);
        if (home_n > 0) {
            const ReduceLoss rl{(const float *)ctx->loss_ws.ptr, loss_stride, n_chunk, B, dense ? t->U + t->I : (int64_t)0, opt->reg, l_parts,
                                l_tickets, loss_per_batch + b0};
            CREC_KLAUNCH(ctx, bpr_call_tail, dim3((unsigned)(nbc * RL_PARTS + home_n * TAIL_COPY_BX)), dim3(RL_THREADS), 0, st, rl,
                         (int)(nbc * RL_PARTS), home_c);
            if (hipPeekAtLastError() == hipSuccess)
                in_shadow = false; // (a launch that failed leaves the copy owed: BringHome runs it on the error return below)
        } else {
            CREC_KLAUNCH(ctx, bpr_reduce_loss, dim3((unsigned)(nbc * RL_PARTS)), dim3(RL_THREADS), 0, st, (const float *)ctx->loss_ws.ptr,
                         loss_stride, n_chunk, B, dense ? t->U + t->I : (int64_t)0, (const int32_t *)w.n_u, (const int32_t *)w.n_q,
                         opt->reg, l_parts, l_tickets, loss_per_batch + b0);
        }
        crec_prof_end(ctx, st);
        if (dp) {
            if (int rc = xdp->allreduce_sum_f32(ctx, loss_per_batch + b0, (size_t)nbc, st))
                return rc;
            // (a transport that waits on its peers inside kernels reports a peer that never came; without draining the stream: a
            // time-out -- after which the tables are no longer the replicas' common state -- is reported by the NEXT chunk's check, the
            // next call's, or crec_comm_status)
            if (int rc = xdp->status_deferred(ctx, st))
                return rc;
        }
        if (piped)
            CREC_HIP(ctx, hipEventRecord(ctx->idx_ev[2 + (ci & 1)], st));
        CREC_HIP(ctx, hipGetLastError());
        if (int rc = crec_prof_flush(ctx, st))
            return rc;
    }
    home.done = true; // an odd number of steps: bring the result home to the caller's buffers
    if (home.run() != (int)hipSuccess)
        return crec_fail(ctx, CREC_EHIP, "copying the tables back from the second buffer set failed");
    return CREC_OK;
}

// ------------------------------------------------------------------------------------
// Row-sharded epoch (include/crec.h "row-sharded training"): users range-sharded, items dealt round-robin, lazy Adam.
// Per chunk of batches: plan (compact ids + request lists) -> counts to the host -> all-to-all-v of the request
// lists (ONE per chunk) -> owner-side index -> the usual occurrence index over the compact ids.
// Per batch: serve rows -> all-to-all-v (rows forward) -> bpr_side_lazy<user> (Adam on the local user rows),
// bpr_side_lazy<item> emitting gradient rows instead of Adam -> all-to-all-v (gradient rows back) -> owner apply.
// ------------------------------------------------------------------------------------
struct ShardTake {
    size_t off = 0;
    char *base = nullptr;
    template <typename T> T *take(size_t n)
    {
        T *p = base ? (T *)(base + off) : nullptr;
        off = align_up(off + sizeof(T) * n, 256);
        return p;
    }
};

int sharded_epoch_impl(crec_ctx *ctx, const crec_bpr_tables *t, int64_t I_global, crec_adam *opt, const int32_t *u,
                       const int32_t *i, const int32_t *j, int64_t N, int32_t B, float *loss_per_batch,
                       crec_shard_stats *stats, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, t && opt, "null argument");
    CREC_REQUIRE(ctx, N >= 1 && u && i && j && loss_per_batch, "the sharded epoch is collective: every rank passes N >= 1 triples");
    CREC_REQUIRE(ctx, t->P && t->Q && t->mP && t->vP && t->mQ && t->vQ, "null table pointer");
    CREC_REQUIRE(ctx, t->d >= 4 && t->d <= 256 && t->d % 4 == 0, "embed_size %d: need a multiple of 4 in [4, 256]", t->d);
    CREC_REQUIRE(ctx, B >= 1 && B <= LAZY_MAX_B, "batch_size %d: batches of 1..%d triples", B, LAZY_MAX_B);
    CREC_REQUIRE(ctx, opt->adam_mode == CREC_ADAM_LAZY, "the row-sharded step is built for adam_mode lazy");
    // batches beyond the in-LDS sorts take the three index builds of a chunk (plan, owner side, compact-id index) through
    // index_big.hip's global-memory sort: the same indices, bit for bit, and no cap on the rows one owner serves per batch
    const bool big = B > MAX_B || ctx->knob_big_index;
    crec_xch *x = crec_xch_of(ctx);
    const int G = x->world, me = x->rank;
    CREC_REQUIRE(ctx, G >= 1 && G <= 64, "world size 1..64");
    const int64_t S = crec_ceil_div(I_global, G); // rows per shard; item g = local row g / G of rank g % G
    CREC_REQUIRE(ctx, I_global >= 1 && (int64_t)G * S < ((int64_t)1 << 31), "item count");
    CREC_REQUIRE(ctx, t->I >= S, "Q shard holds %lld rows, needs ceil(%lld / %d) = %lld", (long long)t->I, (long long)I_global, G, (long long)S);
    CREC_REQUIRE(ctx, t->U >= 1 && t->U < ((int64_t)1 << 31), "table rows");
    const float *tabs[] = {t->P, t->Q, t->mP, t->vP, t->mQ, t->vQ};
    for (const float *p : tabs)
        CREC_REQUIRE(ctx, ((uintptr_t)p & 15) == 0, "table base not 16-byte aligned");
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int d = t->d;
    int lpr = 1;
    while (lpr * 4 < d)
        lpr <<= 1;
    const int64_t nb = crec_ceil_div(N, B);
    const int64_t IC = 2 * (int64_t)B; // compact item id space of a batch
    StepFn step = pick_step(d, false);
    const int CH = ctx->knob_ch > 0 ? ctx->knob_ch : 16;
    CREC_REQUIRE(ctx, CH >= 1 && CH <= 255, "option ch: 1..255");

    const size_t per_batch = layout_index(nullptr, 1, B, t->U, IC, false, CH).bytes;
    int64_t chunk = (int64_t)((size_t)1 << 30) / (int64_t)per_batch;
    chunk = chunk < 1 ? 1 : (chunk > 1024 ? 1024 : chunk);
    if (big && chunk * B > ((int64_t)1 << 22)) // index_big.hip's scratch: ~100 bytes per triple of the chunk
        chunk = ((int64_t)1 << 22) / B;
    chunk = chunk < 1 ? 1 : chunk;
    chunk = chunk > nb ? nb : chunk;
    // a transport that stages a rank's payload (the peer windows) bounds what one exchange carries: per step the rows of a batch,
    // per chunk the request lists of its batches (every rank holds the same B, d and window, so every rank cuts the same chunks)
    CREC_REQUIRE(ctx, IC * d * 4 + 16 * G <= x->max_bytes(), "the transport's window is too small for a batch's rows (%lld bytes)", (long long)IC * d * 4);
    if (chunk * (IC * 4 + 4 * G + 16) + 16 * G > x->max_bytes())
        chunk = (x->max_bytes() - 16 * G) / (IC * 4 + 4 * G + 16);
    chunk = chunk < 1 ? 1 : chunk;
    if (int rc = crec_reserve(ctx, &ctx->idx_ws, layout_index(nullptr, chunk, B, t->U, IC, false, CH).bytes))
        return rc;
    const int64_t loss_stride = 2 * (int64_t)B; // x_b, then the l2 term of every triple
    if (int rc = crec_reserve(ctx, &ctx->loss_ws, sizeof(float) * (size_t)chunk * loss_stride))
        return rc;
    const IndexWs w0 = layout_index(nullptr, 1, B, t->U, IC, false, CH);
    const size_t part_bytes = align_up(sizeof(float) * (size_t)(w0.HIu + w0.HIq) * d, 256);
    const size_t ticket_bytes = sizeof(int32_t) * (size_t)(w0.HRu + w0.HRq);
    if (int rc = crec_reserve(ctx, &ctx->misc, part_bytes + ticket_bytes))
        return rc;
    // the finishing wave of a heavy row re-zeroes its counter, so the counters only need clearing when they move
    if (ticket_bytes && (ctx->tickets_at != (char *)ctx->misc.ptr + part_bytes || ctx->tickets_bytes != ticket_bytes)) {
        CREC_HIP(ctx, hipMemsetAsync((char *)ctx->misc.ptr + part_bytes, 0, ticket_bytes, st));
        ctx->tickets_at = (char *)ctx->misc.ptr + part_bytes;
        ctx->tickets_bytes = ticket_bytes;
    }


    // plan-side workspace (sizes known up front) ...
    struct PlanWs {
        int32_t *req_keys, *n_req, *req_off, *ci, *cj, *cnt_send, *cnt_recv, *srv_cnt, *ids_send;
        int64_t *pack_off, *srv_base;
        float *Qc, *dQc;
        size_t bytes;
    };
    const auto plan_layout = [&](void *base, int64_t nbc) {
        ShardTake k;
        k.base = (char *)base;
        PlanWs p{};
        p.req_keys = k.take<int32_t>((size_t)nbc * IC);
        p.n_req = k.take<int32_t>((size_t)nbc);
        p.req_off = k.take<int32_t>((size_t)nbc * (G + 1));
        p.ci = k.take<int32_t>((size_t)nbc * B);
        p.cj = k.take<int32_t>((size_t)nbc * B);
        p.cnt_send = k.take<int32_t>((size_t)nbc * G);
        p.cnt_recv = k.take<int32_t>((size_t)nbc * G);
        p.srv_cnt = k.take<int32_t>((size_t)nbc * G);
        p.ids_send = k.take<int32_t>((size_t)nbc * IC);
        p.pack_off = k.take<int64_t>((size_t)nbc * G);
        p.srv_base = k.take<int64_t>((size_t)nbc * G);
        p.Qc = k.take<float>((size_t)IC * d);
        p.dQc = k.take<float>((size_t)IC * d);
        p.bytes = k.off;
        return p;
    };
    if (int rc = crec_reserve(ctx, &ctx->shard_ws, plan_layout(nullptr, chunk).bytes))
        return rc;
    // ... and the serve side, sized once the peers' request counts are known
    struct ServeWs {
        int32_t *ids_recv, *srv_rows, *order, *urows, *useg, *n_urows;
        float *sendrows, *recvgrads;
        size_t bytes;
    };
    const auto serve_layout = [&](void *base, int64_t nbc, int64_t total_ids, int64_t stride) {
        ShardTake k;
        k.base = (char *)base;
        ServeWs v{};
        v.ids_recv = k.take<int32_t>((size_t)total_ids + 1);
        v.srv_rows = k.take<int32_t>((size_t)nbc * stride);
        v.order = k.take<int32_t>((size_t)nbc * stride);
        v.urows = k.take<int32_t>((size_t)nbc * stride);
        v.useg = k.take<int32_t>((size_t)nbc * (stride + 1));
        v.n_urows = k.take<int32_t>((size_t)nbc);
        v.sendrows = k.take<float>((size_t)stride * d);
        v.recvgrads = k.take<float>((size_t)stride * d);
        v.bytes = k.off;
        return v;
    };
    // host staging (page-locked: the per-chunk counts cross PCIe a few times)
    const size_t host_ints = (size_t)chunk * (G + 1) + 4 * (size_t)chunk * G + 8 * (size_t)chunk + 64;
    if (host_ints > ctx->pinned64_n) {
        if (ctx->pinned64)
            CREC_HIP(ctx, hipHostFree(ctx->pinned64));
        ctx->pinned64 = nullptr;
        ctx->pinned64_n = 0;
        CREC_HIP(ctx, hipHostMalloc((void **)&ctx->pinned64, sizeof(int64_t) * host_ints * 2, hipHostMallocDefault));
        ctx->pinned64_n = host_ints * 2;
    }

    const int cap_plan = (int)((IC + 63) / 64 * 64);
    if (!big)
        CREC_HIP(ctx, hipFuncSetAttribute((const void *)shard_plan_batches, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)shard_lds_bytes(cap_plan)));
    const int cap_idx = (2 * B + 63) / 64 * 64;
    const size_t lds_idx = big ? 0 : index_lds_bytes(cap_idx);
    if (lds_idx > ctx->index_lds_set) {
        CREC_HIP(ctx, hipFuncSetAttribute((const void *)bpr_index_batches, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_idx));
        ctx->index_lds_set = lds_idx;
    }

    StepArgs a{};
    a.ctx = ctx;
    a.d = d;
    a.U = t->U;
    a.I = IC;
    a.B = B;
    a.st = st;
    a.partial = (float *)ctx->misc.ptr;
    a.tickets = (int32_t *)((char *)ctx->misc.ptr + part_bytes);
    if (int rc = setup_lazy_bufs(ctx, B, d, CH, st, &a.lazy))
        return rc;
    int64_t sent_ids = 0, sent_rows = 0, sent_grads = 0, n_requested = 0, n_served = 0, max_served = 0;
    const bool fused = x->next_half() != nullptr && !ctx->knob_shard_unfused; // the peer transport: exchanges fused with serve / apply

    std::vector<int64_t> scnt(G), sdsp(G), rcnt(G), rdsp(G);
    for (int64_t b0 = 0; b0 < nb; b0 += chunk) {
        const crec_range chunk_range("crec_bpr chunk of batches");
        const int64_t nbc = (nb - b0) < chunk ? (nb - b0) : chunk;
        const int64_t t0 = b0 * B, n_chunk = (N - t0) < nbc * B ? (N - t0) : nbc * B;
        const PlanWs p = plan_layout(ctx->shard_ws.ptr, nbc);
        // 1. plan: compact ids and request lists of every batch of the chunk
        if (big) {
            if (int rc = crec_shard_plan_big(ctx, i + t0, j + t0, n_chunk, B, nbc, G, S, row_bits((int64_t)G * S), p.req_keys, p.n_req, p.req_off,
                                             p.ci, p.cj, st))
                return rc;
        } else {
            crec_prof_begin(ctx, CREC_K_SHARD_PLAN, st);
            CREC_KLAUNCH(ctx, shard_plan_batches, dim3((unsigned)nbc), dim3(SHARD_THREADS), shard_lds_bytes(cap_plan), st, i + t0,
                         j + t0, n_chunk, B, cap_plan, G, S, row_bits((int64_t)G * S), p.req_keys, p.n_req, p.req_off, p.ci, p.cj);
            crec_prof_end(ctx, st);
        }
        // 2. the per-owner counts -> host; every rank tells every owner how many rows each batch will ask for
        char *hp = (char *)ctx->pinned64;
        const auto htake = [&hp](size_t bytes) {
            char *q = hp;
            hp += (bytes + 7) / 8 * 8;
            return q;
        };
        int32_t *h_off = (int32_t *)htake(sizeof(int32_t) * (size_t)nbc * (G + 1)); // [nbc][G+1]
        int32_t *h_cnt_send = (int32_t *)htake(sizeof(int32_t) * (size_t)nbc * G);  // [G][nbc]
        int32_t *h_cnt_recv = (int32_t *)htake(sizeof(int32_t) * (size_t)nbc * G);  // [G][nbc]  rows peer p asks of me in batch b
        int32_t *h_srv_cnt = (int32_t *)htake(sizeof(int32_t) * (size_t)nbc * G);   // [nbc][G]
        int64_t *h_pack = (int64_t *)htake(sizeof(int64_t) * (size_t)nbc * G);      // [nbc][G]
        int64_t *h_base = (int64_t *)htake(sizeof(int64_t) * (size_t)nbc * G);      // [nbc][G]
        float *h_verdict = (float *)htake(64);                                         // (page-locked like the rest: see crec_peer.hip, err_host)
        CREC_HIP(ctx, hipMemcpyAsync(h_off, p.req_off, sizeof(int32_t) * (size_t)nbc * (G + 1), hipMemcpyDeviceToHost, st));
        CREC_HIP(ctx, hipStreamSynchronize(st));
        for (int o = 0; o < G; ++o)
            for (int64_t b = 0; b < nbc; ++b)
                h_cnt_send[(size_t)o * nbc + b] = h_off[b * (G + 1) + o + 1] - h_off[b * (G + 1) + o];
        CREC_HIP(ctx, hipMemcpyAsync(p.cnt_send, h_cnt_send, sizeof(int32_t) * (size_t)nbc * G, hipMemcpyHostToDevice, st));
        for (int q = 0; q < G; ++q) {
            scnt[q] = rcnt[q] = nbc;
            sdsp[q] = rdsp[q] = (int64_t)q * nbc;
        }
        if (int rc = x->alltoallv(ctx, p.cnt_send, scnt.data(), sdsp.data(), p.cnt_recv, rcnt.data(), rdsp.data(), 4, st))
            return rc;
        CREC_HIP(ctx, hipMemcpyAsync(h_cnt_recv, p.cnt_recv, sizeof(int32_t) * (size_t)nbc * G, hipMemcpyDeviceToHost, st));
        if (int rc = x->status(ctx, st)) // the counts steer the host from here on: a peer that never delivered them is an error now
            return rc;
        CREC_HIP(ctx, hipStreamSynchronize(st));
        // 3. where everything goes: owner-major packing of my requests, peer-major layout of what I serve
        int64_t at = 0, total_srv = 0, stride = 64;
        for (int o = 0; o < G; ++o) {
            sdsp[o] = at;
            for (int64_t b = 0; b < nbc; ++b) {
                h_pack[b * G + o] = at;
                at += h_cnt_send[(size_t)o * nbc + b];
            }
            scnt[o] = at - sdsp[o];
        }
        n_requested += at;
        for (int q = 0; q < G; ++q) {
            rdsp[q] = total_srv;
            for (int64_t b = 0; b < nbc; ++b) {
                h_base[b * G + q] = total_srv;
                h_srv_cnt[b * G + q] = h_cnt_recv[(size_t)q * nbc + b];
                total_srv += h_cnt_recv[(size_t)q * nbc + b];
            }
            rcnt[q] = total_srv - rdsp[q];
        }
        n_served += total_srv;
        for (int64_t b = 0; b < nbc; ++b) {
            int64_t n_b = 0;
            for (int q = 0; q < G; ++q)
                n_b += h_srv_cnt[b * G + q];
            stride = n_b > stride ? n_b : stride;
        }
        max_served = stride > max_served ? stride : max_served;
        stride = (stride + 63) / 64 * 64;
        // What this rank is asked for is data-dependent and differs from rank to rank: a rank that left here alone would leave its
        // peers blocked in the exchanges below for ever.  So the verdict is collective: every rank contributes whether IT can go
        // on (one float through the all-reduce the chunk ends with anyway) and all of them fail together.
        int local_rc = CREC_OK;
        if (!big && stride > 16 * SHARD_THREADS)
            local_rc = crec_fail(ctx, CREC_EINVAL, "a batch asks this rank for %lld rows; the owner-side index sorts at most %d per batch in LDS",
                                 (long long)stride, 16 * SHARD_THREADS);
        else
            local_rc = crec_reserve(ctx, &ctx->shard_ws2, serve_layout(nullptr, nbc, total_srv, stride).bytes);
        if (big && local_rc == CREC_OK) // (the owner-side index' scratch is sized by what THIS rank is asked for: part of the same verdict)
            local_rc = crec_shard_owner_index_big_reserve(ctx, nbc, stride, st);
        // ... and what it is asked for must fit a half of the transport's window: the rows it serves are staged there (next_half) and the
        // gradient rows it receives travel through it.  n_srv reaches G * IC under skew, beyond the IC * d * 4 checked up front; a rank
        // that found out alone -- inside alltoallv_inplace, after shard_serve_rows had already written past the half its peers are still
        // pulling from -- would leave them waiting for the time-out
        if (local_rc == CREC_OK && stride * d * 4 + 16 * G > x->max_bytes())
            local_rc = crec_fail(ctx, CREC_EINVAL, "a batch asks this rank for %lld rows = %lld bytes; a half of the transport's window holds %lld",
                                 (long long)stride, (long long)stride * d * 4, (long long)x->max_bytes());
        if (G > 1) {
            if (int rc = crec_reserve(ctx, &ctx->shard_flag, 64))
                local_rc = local_rc ? local_rc : rc;
            float verdict = local_rc == CREC_OK ? 0.f : 1.f;
            if (ctx->shard_flag.ptr) {
                *h_verdict = verdict;
                CREC_HIP(ctx, hipMemcpyAsync(ctx->shard_flag.ptr, h_verdict, sizeof(float), hipMemcpyHostToDevice, st));
                if (int rc = x->allreduce_sum_f32(ctx, (float *)ctx->shard_flag.ptr, 1, st))
                    return local_rc ? local_rc : rc;
                CREC_HIP(ctx, hipMemcpyAsync(h_verdict, ctx->shard_flag.ptr, sizeof(float), hipMemcpyDeviceToHost, st));
                CREC_HIP(ctx, hipStreamSynchronize(st));
                verdict = *h_verdict;
            }
            if (local_rc == CREC_OK && verdict > 0.f)
                return crec_fail(ctx, CREC_EINVAL, "%d peer rank(s) cannot serve this chunk (more rows asked of one owner than its index sorts per batch "
                                                   "or than a half of its transport window holds, or out of memory there): every rank stops here", (int)(verdict + 0.5f));
        }
        if (local_rc)
            return local_rc;
        const ServeWs v = serve_layout(ctx->shard_ws2.ptr, nbc, total_srv, stride);
        CREC_HIP(ctx, hipMemcpyAsync(p.pack_off, h_pack, sizeof(int64_t) * (size_t)nbc * G, hipMemcpyHostToDevice, st));
        CREC_HIP(ctx, hipMemcpyAsync(p.srv_base, h_base, sizeof(int64_t) * (size_t)nbc * G, hipMemcpyHostToDevice, st));
        CREC_HIP(ctx, hipMemcpyAsync(p.srv_cnt, h_srv_cnt, sizeof(int32_t) * (size_t)nbc * G, hipMemcpyHostToDevice, st));
        // 4. the index exchange, once per chunk
        hipLaunchKernelGGL(shard_pack_ids, dim3((unsigned)nbc), dim3(256), 0, st, (const int32_t *)p.req_keys,
                           (const int32_t *)p.req_off, (const int64_t *)p.pack_off, B, G, S, p.ids_send);
        if (int rc = x->alltoallv(ctx, p.ids_send, scnt.data(), sdsp.data(), v.ids_recv, rcnt.data(), rdsp.data(), 4, st))
            return rc;
        for (int q = 0; q < G; ++q)
            sent_ids += q == me ? 0 : 4 * (scnt[q] + nbc);
        // the request lists are row ids the kernels below index Q with: a peer that never delivered them (a transport that waits inside
        // its kernels gives up after its time-out and leaves the receive buffer as it was) must be an error here, not an address
        if (int rc = x->status(ctx, st))
            return rc;
        // 5. owner-side index of every batch, then the occurrence index over the compact ids
        const int cap_srv = (int)stride;
        IndexWs w = layout_index(ctx->idx_ws.ptr, nbc, B, t->U, IC, false, CH, 0, 0, lazy_spw(d));
        if (big) {
            if (int rc = crec_shard_owner_index_big(ctx, v.ids_recv, p.srv_base, p.srv_cnt, G, nbc, stride, row_bits(S), v.srv_rows, v.order, v.urows,
                                                    v.useg, v.n_urows, st))
                return rc;
            const LazyIndexOut o{w.u_desc, w.i_desc, w.j_desc, w.i_tmp, w.j_tmp, w.slot_pu, w.order, w.nslots};
            if (int rc = crec_lazy_index_big(ctx, u + t0, p.ci, p.cj, n_chunk, B, nbc, w.bits_u, w.bits_q, w.spw, o, st))
                return rc;
        } else {
            CREC_HIP(ctx, hipFuncSetAttribute((const void *)shard_owner_index, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)shard_lds_bytes(cap_srv)));
            hipLaunchKernelGGL(shard_owner_index, dim3((unsigned)nbc), dim3(SHARD_THREADS), shard_lds_bytes(cap_srv), st,
                               (const int32_t *)v.ids_recv, (const int64_t *)p.srv_base, (const int32_t *)p.srv_cnt, G, cap_srv,
                               cap_srv, row_bits(S), v.srv_rows, v.order, v.urows, v.useg, v.n_urows);
            crec_prof_begin(ctx, CREC_K_INDEX, st);
            CREC_KLAUNCH(ctx, bpr_index_batches, dim3((unsigned)(2 * nbc)), dim3(PREP_THREADS), lds_idx, st, u + t0,
                         (const int32_t *)p.ci, (const int32_t *)p.cj, n_chunk, B, cap_idx, w, 1);
            crec_prof_end(ctx, st);
        }
        const int32_t *hn = nullptr; // the lazy step's launches are sized by the batch alone
        a.cur = Tables{t->P, t->mP, t->vP, p.Qc, nullptr, nullptr};
        // 6. the steps
        for (int64_t b = 0; b < nbc; ++b) {
            int64_t n_srv = 0;
            for (int q = 0; q < G; ++q) { // forward: I send peer q the rows it asked for, owner o sends me mine
                scnt[q] = (int64_t)h_srv_cnt[b * G + q] * d;
                sdsp[q] = n_srv * d;
                n_srv += h_srv_cnt[b * G + q];
                rcnt[q] = (int64_t)(h_off[b * (G + 1) + q + 1] - h_off[b * (G + 1) + q]) * d;
                rdsp[q] = (int64_t)h_off[b * (G + 1) + q] * d;
                if (q != me) {
                    sent_rows += 4 * scnt[q];
                    sent_grads += 4 * rcnt[q];
                }
            }
            if (fused) { // peer transport: the rows are served straight into this rank's window, the exchange starts from there
                if (n_srv > 0) {
                    crec_prof_begin(ctx, CREC_K_SHARD_SERVE, st);
                    CREC_KLAUNCH(ctx, shard_serve_rows, dim3((unsigned)crec_ceil_div(n_srv * lpr, 256)), dim3(256), 0, st,
                                 (const float *)t->Q, (const int32_t *)(v.srv_rows + b * stride), (int)n_srv, d, lpr, (float *)x->next_half(), 1);
                    crec_prof_end(ctx, st);
                }
                if (int rc = x->alltoallv_inplace(ctx, scnt.data(), sdsp.data(), p.Qc, rcnt.data(), rdsp.data(), st))
                    return rc;
            } else {
                if (n_srv > 0) {
                    crec_prof_begin(ctx, CREC_K_SHARD_SERVE, st);
                    CREC_KLAUNCH(ctx, shard_serve_rows, dim3((unsigned)crec_ceil_div(n_srv * lpr, 256)), dim3(256), 0, st,
                                 (const float *)t->Q, (const int32_t *)(v.srv_rows + b * stride), (int)n_srv, d, lpr, v.sendrows, 0);
                    crec_prof_end(ctx, st);
                }
                if (int rc = x->alltoallv(ctx, v.sendrows, scnt.data(), sdsp.data(), p.Qc, rcnt.data(), rdsp.data(), 4, st))
                    return rc;
            }
            a.h.opt = 0;
            a.h.lr_tab = nullptr;
            a.h.lr_t = (opt->lr * sqrtf(1.0f - opt->beta2_power)) / (1.0f - opt->beta1_power);
            a.h.beta1 = opt->beta1;
            a.h.beta2 = opt->beta2;
            a.h.eps = opt->eps;
            a.h.reg = opt->reg;
            // the item side emits each compact row's gradient sum (incl. its l2 term) instead of Adam -- over the peer transport straight
            // into this rank's window, where the owners will read it (the layout of the compact table: owner o's rows from req_off[o])
            a.h.dQ = fused ? (float *)x->next_half() : p.dQc;
            a.h.dq_sys = fused;
            a.xb = (float *)ctx->loss_ws.ptr + b * loss_stride;
            a.nb_cur = (int32_t)((n_chunk - b * B) < B ? (n_chunk - b * B) : B);
            a.rowterm = a.xb + B;
            step(a, w, b, hn + b * HN);
            // back: the gradient row of every item I referenced goes to its owner (the forward sizes, reversed)
            if (fused) { // exchange + owner apply in one launch: the owners read the gradient rows where the step left them
                crec_peer_view pv;
                CREC_REQUIRE(ctx, x->begin(&pv), "peer transport: not connected");
                ShardPeerOffs so{};
                for (int q = 0; q < G; ++q) {
                    so.seg_off[q] = rdsp[q] * 4;
                    so.slot_base[q] = (int32_t)(sdsp[q] / d);
                }
                so.slot_base[G] = (int32_t)n_srv;
                int64_t nblk = crec_ceil_div(n_srv * lpr, PEER_THREADS);
                // (no copy-in here: workgroup 0 publishes at once, so no workgroup of this launch waits for another one of it and the
                // grid may exceed what is resident -- a lane group per row, as in shard_owner_apply)
                nblk = nblk < 1 ? 1 : (nblk > 8 * PEER_MAX_BLOCKS ? 8 * PEER_MAX_BLOCKS : nblk);
                crec_prof_begin(ctx, CREC_K_SHARD_APPLY, st);
                CREC_KLAUNCH(ctx, shard_exchange_apply, dim3((unsigned)nblk), dim3(PEER_THREADS), 0, st, pv, so, t->Q, t->mQ, t->vQ,
                             (const int32_t *)(v.urows + b * stride), (const int32_t *)(v.useg + b * (stride + 1)),
                             (const int32_t *)(v.order + b * stride), (const int32_t *)(v.n_urows + b), d, lpr, a.h);
                crec_prof_end(ctx, st);
            } else {
                if (int rc = x->alltoallv(ctx, p.dQc, rcnt.data(), rdsp.data(), v.recvgrads, scnt.data(), sdsp.data(), 4, st))
                    return rc;
                if (n_srv > 0) {
                    crec_prof_begin(ctx, CREC_K_SHARD_APPLY, st);
                    CREC_KLAUNCH(ctx, shard_owner_apply, dim3((unsigned)crec_ceil_div(n_srv * lpr, 256)), dim3(256), 0, st, t->Q, t->mQ,
                                 t->vQ, (const int32_t *)(v.urows + b * stride), (const int32_t *)(v.useg + b * (stride + 1)),
                                 (const int32_t *)(v.order + b * stride), (const int32_t *)(v.n_urows + b),
                                 (const float *)v.recvgrads, d, lpr, a.h);
                    crec_prof_end(ctx, st);
                }
            }
            opt->beta1_power = opt->beta1_power * opt->beta1;
            opt->beta2_power = opt->beta2_power * opt->beta2;
            opt->step += 1;
        }
        double *l_parts;
        int32_t *l_tickets;
        if (int rc = loss_parts(ctx, chunk, st, &l_parts, &l_tickets))
            return rc;
        crec_prof_begin(ctx, CREC_K_REDUCE_LOSS, st);
        CREC_KLAUNCH(ctx, bpr_reduce_loss, dim3((unsigned)(nbc * RL_PARTS)), dim3(RL_THREADS), 0, st, (const float *)ctx->loss_ws.ptr, loss_stride,
                     n_chunk, B, (int64_t)0, (const int32_t *)w.n_u, (const int32_t *)w.n_q, opt->reg, l_parts, l_tickets, loss_per_batch + b0);
        crec_prof_end(ctx, st);
        if (int rc = x->allreduce_sum_f32(ctx, loss_per_batch + b0, (size_t)nbc, st))
            return rc;
        if (int rc = x->status(ctx, st))
            return rc;
        CREC_HIP(ctx, hipGetLastError());
        if (int rc = crec_prof_flush(ctx, st))
            return rc;
    }
    if (stats) {
        stats->bytes_ids_sent += sent_ids;
        stats->bytes_rows_sent += sent_rows;
        stats->bytes_grads_sent += sent_grads;
        stats->rows_requested += n_requested;
        stats->rows_served += n_served;
        stats->max_rows_served_per_batch = max_served > stats->max_rows_served_per_batch ? max_served : stats->max_rows_served_per_batch;
        stats->steps += nb;
    }
    return CREC_OK;
}
} // namespace

extern "C" {

int crec_bpr_train_epoch(crec_ctx *ctx, const crec_bpr_tables *t, crec_adam *opt, const int32_t *u, const int32_t *i,
                         const int32_t *j, int64_t N, int32_t B, float *loss_per_batch, void *stream)
{
    // tf1 beyond the in-LDS index (model/Recommender.py:17 takes any batch_size): the call runs as deferred Adam -- every row current
    // at the call's first step, the touched rows replaying the steps they sat out -- and ends with the flush, so the caller's tables
    // are TF1's after every call, as they are after every sess.run (one whole-table sweep per call instead of one per step)
    // The same form where the sweep itself is what a step costs: tables whose rows and slots make 64 MB and more (24 d (U + I) bytes
    // per step at the memory system's rate: 12 us upwards, against the 14-18 us of a touched-rows step whatever the tables' size).
    // us per step, swept / so (tools/time_tf1_vs_deferred.py, d = 64): 20 k x 10 k rows 19.9 / 19.0, 50 k x 20 k 30.2 / 23.2,
    // 138 k x 27 k 56.5 / 30.5 (d = 128: 93.8 / 36.5), 500 k x 100 k 172.7 / 47.3.  Option "tf1_sweep" = 1 keeps the sweep.
    const bool big_tables = ctx && t && 24.0 * t->d * ((double)t->U + (double)t->I) >= 64.0 * 1024 * 1024 && !ctx->knob_tf1_sweep;
    if (ctx && t && opt && opt->adam_mode == CREC_ADAM_TF1 && (B > MAX_B || big_tables) && B <= LAZY_MAX_B && opt->beta1 <= 0.9000001f && N > 0 &&
        t->U > 0 && t->I > 0 && ctx->device >= 0) {
        CREC_HIP(ctx, hipSetDevice(ctx->device));
        hipStream_t st = (hipStream_t)stream;
        if (int rc = crec_reserve(ctx, &ctx->tf1_last, sizeof(int32_t) * ((size_t)t->U + (size_t)t->I)))
            return rc;
        crec_bpr_deferred df{(int32_t *)ctx->tf1_last.ptr, (int32_t *)ctx->tf1_last.ptr + t->U};
        CREC_HIP(ctx, hipMemsetD32Async((hipDeviceptr_t)ctx->tf1_last.ptr, (int)opt->step, (size_t)t->U + (size_t)t->I, st));
        crec_adam o2 = *opt;
        o2.adam_mode = CREC_ADAM_DEFERRED;
        if (int rc = train_epoch_impl(ctx, t, &o2, u, i, j, N, B, loss_per_batch, stream, false, &df))
            return rc;
        if (int rc = crec_bpr_adam_flush(ctx, t, &df, &o2, stream))
            return rc;
        o2.adam_mode = opt->adam_mode;
        *opt = o2;
        return CREC_OK;
    }
    return train_epoch_impl(ctx, t, opt, u, i, j, N, B, loss_per_batch, stream, false);
}

int crec_bpr_train_epoch_dp(crec_ctx *ctx, const crec_bpr_tables *t, crec_adam *opt, const int32_t *u,
                            const int32_t *i, const int32_t *j, int64_t N, int32_t B, float *loss_per_batch,
                            void *stream)
{
    return train_epoch_impl(ctx, t, opt, u, i, j, N, B, loss_per_batch, stream, true);
}

int crec_bpr_train_epoch_deferred(crec_ctx *ctx, const crec_bpr_tables *t, const crec_bpr_deferred *df, crec_adam *opt,
                                  const int32_t *u, const int32_t *i, const int32_t *j, int64_t N, int32_t B,
                                  float *loss_per_batch, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, df != nullptr, "null argument");
    // the replay of the steps a row sat out stops after CREC_REPLAY_MAX steps, where m has left the fp32 range -- true for beta1 <= 0.9
    // (0.9^1024 = 1e-47); a slower decay would still carry weight there and the mode would no longer equal the TF1 sweep
    CREC_REQUIRE(ctx, opt && opt->beta1 <= 0.9000001f, "adam_mode deferred replays at most %d idle steps exactly: beta1 = %g > 0.9 decays too slowly for that (use tf1 or lazy)",
                 CREC_REPLAY_MAX, opt ? (double)opt->beta1 : 0.0);
    return train_epoch_impl(ctx, t, opt, u, i, j, N, B, loss_per_batch, stream, false, df);
}

int crec_bpr_adam_flush(crec_ctx *ctx, const crec_bpr_tables *t, const crec_bpr_deferred *df, const crec_adam *opt, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, t && df && opt && df->lastP && df->lastQ, "null argument");
    CREC_REQUIRE(ctx, opt->adam_mode == CREC_ADAM_DEFERRED, "crec_bpr_adam_flush belongs to adam_mode deferred");
    CREC_REQUIRE(ctx, t->d >= 4 && t->d <= 256 && t->d % 4 == 0, "embed_size");
    if (opt->step == 0)
        return CREC_OK;
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    Hyper h{};
    if (int rc = lr_history(ctx, opt, opt->step, st, &h.lr_tab))
        return rc;
    h.beta1 = opt->beta1;
    h.beta2 = opt->beta2;
    h.eps = opt->eps;
    h.step = (int32_t)opt->step;
    int lpr = 1;
    while (lpr * 4 < t->d)
        lpr <<= 1;
    hipLaunchKernelGGL(bpr_adam_flush, dim3((unsigned)crec_ceil_div(t->U * lpr, 256)), dim3(256), 0, st, t->P, t->mP, t->vP, df->lastP,
                       t->U, (int)t->d, lpr, h);
    hipLaunchKernelGGL(bpr_adam_flush, dim3((unsigned)crec_ceil_div(t->I * lpr, 256)), dim3(256), 0, st, t->Q, t->mQ, t->vQ, df->lastQ,
                       t->I, (int)t->d, lpr, h);
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}

int crec_bpr_train_epoch_sharded(crec_ctx *ctx, const crec_bpr_tables *t, int64_t I_global, crec_adam *opt, const int32_t *u,
                                 const int32_t *i, const int32_t *j, int64_t N, int32_t B, float *loss_per_batch,
                                 crec_shard_stats *stats, void *stream)
{
    return sharded_epoch_impl(ctx, t, I_global, opt, u, i, j, N, B, loss_per_batch, stats, stream);
}

// test hooks: the index kernels of the row-sharded step on ONE batch, to be held against their host form (shard_host.cpp).
// All array arguments are device pointers except srv_cnt (host).
int crec_debug_shard_plan(crec_ctx *ctx, const int32_t *i, const int32_t *j, int32_t nb, int32_t G, int64_t I_global, int32_t *req_keys,
                          int32_t *n_req, int32_t *req_off, int32_t *ci, int32_t *cj, int32_t *ids_send, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, i && j && req_keys && n_req && req_off && ci && cj && ids_send, "null argument");
    const bool big = nb > MAX_B || ctx->knob_big_index; // (the form batches beyond the in-LDS sort take: index_big.hip)
    CREC_REQUIRE(ctx, nb >= 1 && nb <= LAZY_MAX_B && G >= 1 && G <= 64 && I_global >= 1, "shape");
    const int64_t S = crec_ceil_div(I_global, G);
    CREC_REQUIRE(ctx, (int64_t)G * S < ((int64_t)1 << 31), "item count");
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int cap = (2 * nb + 63) / 64 * 64;
    if (big) {
        if (int rc = crec_shard_plan_big(ctx, i, j, (int64_t)nb, nb, 1, G, S, row_bits((int64_t)G * S), req_keys, n_req, req_off, ci, cj, st))
            return rc;
    } else {
        CREC_HIP(ctx, hipFuncSetAttribute((const void *)shard_plan_batches, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shard_lds_bytes(cap)));
        hipLaunchKernelGGL(shard_plan_batches, dim3(1), dim3(SHARD_THREADS), shard_lds_bytes(cap), st, i, j, (int64_t)nb, nb, cap, G, S,
                           row_bits((int64_t)G * S), req_keys, n_req, req_off, ci, cj);
    }
    std::vector<int32_t> off((size_t)G + 1);
    CREC_HIP(ctx, hipMemcpyAsync(off.data(), req_off, sizeof(int32_t) * ((size_t)G + 1), hipMemcpyDeviceToHost, st));
    CREC_HIP(ctx, hipStreamSynchronize(st));
    std::vector<int64_t> pack((size_t)G);
    for (int o = 0; o < G; ++o)
        pack[(size_t)o] = off[(size_t)o]; // one batch: owner o's requests start where its keys start
    int64_t *d_pack = nullptr;
    CREC_HIP(ctx, hipMalloc((void **)&d_pack, sizeof(int64_t) * (size_t)G));
    hipError_t e = hipMemcpyAsync(d_pack, pack.data(), sizeof(int64_t) * (size_t)G, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(shard_pack_ids, dim3(1), dim3(256), 0, st, (const int32_t *)req_keys, (const int32_t *)req_off,
                           (const int64_t *)d_pack, nb, G, S, ids_send);
        e = hipStreamSynchronize(st);
    }
    (void)hipFree(d_pack);
    CREC_HIP(ctx, e);
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}

int crec_debug_shard_owner_index(crec_ctx *ctx, const int32_t *ids_recv, const int32_t *srv_cnt, int32_t G, int64_t S, int32_t *order,
                                 int32_t *urows, int32_t *useg, int32_t *n_urows, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, ids_recv && srv_cnt && order && urows && useg && n_urows, "null argument");
    CREC_REQUIRE(ctx, G >= 1 && G <= 64 && S >= 1 && S < ((int64_t)1 << 31), "shape");
    std::vector<int64_t> base((size_t)G);
    int64_t n = 0;
    for (int p = 0; p < G; ++p) {
        CREC_REQUIRE(ctx, srv_cnt[p] >= 0, "negative count");
        base[(size_t)p] = n;
        n += srv_cnt[p];
    }
    const int64_t stride = (n + 63) / 64 * 64 > 64 ? (n + 63) / 64 * 64 : 64;
    const bool big = stride > 16 * SHARD_THREADS || ctx->knob_big_index;
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    char *tmp = nullptr;
    const size_t bytes = 256 + sizeof(int64_t) * (size_t)G + 256 + sizeof(int32_t) * (size_t)G + 256 + sizeof(int32_t) * (size_t)stride;
    CREC_HIP(ctx, hipMalloc((void **)&tmp, bytes));
    int64_t *d_base = (int64_t *)tmp;
    int32_t *d_cnt = (int32_t *)(tmp + align_up(sizeof(int64_t) * (size_t)G, 256));
    int32_t *d_rows = (int32_t *)((char *)d_cnt + align_up(sizeof(int32_t) * (size_t)G, 256));
    hipError_t e = hipMemcpyAsync(d_base, base.data(), sizeof(int64_t) * (size_t)G, hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_cnt, srv_cnt, sizeof(int32_t) * (size_t)G, hipMemcpyHostToDevice, st);
    int big_rc = CREC_OK;
    if (e == hipSuccess && big) {
        big_rc = crec_shard_owner_index_big(ctx, ids_recv, (const int64_t *)d_base, (const int32_t *)d_cnt, G, 1, stride, row_bits(S), d_rows, order, urows,
                                            useg, n_urows, st);
        e = hipStreamSynchronize(st);
    } else {
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void *)shard_owner_index, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shard_lds_bytes((int)stride));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(shard_owner_index, dim3(1), dim3(SHARD_THREADS), shard_lds_bytes((int)stride), st, ids_recv,
                               (const int64_t *)d_base, (const int32_t *)d_cnt, G, (int)stride, (int)stride, row_bits(S), d_rows, order, urows,
                               useg, n_urows);
            e = hipStreamSynchronize(st);
        }
    }
    (void)hipFree(tmp);
    if (big_rc)
        return big_rc;
    CREC_HIP(ctx, e);
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}

int crec_debug_lazy_index(crec_ctx *ctx, const int32_t *u, const int32_t *i, const int32_t *j, int64_t N, int32_t B, int64_t U, int64_t I,
                          int32_t d, int32_t big, int32_t *u_desc, int32_t *i_desc, int32_t *j_desc, int32_t *slot_pu, int32_t *nslots,
                          void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, u && i && j && u_desc && i_desc && j_desc && slot_pu && nslots && N >= 1, "null argument");
    CREC_REQUIRE(ctx, B >= 1 && (big ? B <= LAZY_MAX_B : B <= MAX_B) && U >= 1 && I >= 1 && d >= 4 && d <= 256, "shape");
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t nbc = crec_ceil_div(N, B);
    if (int rc = crec_reserve(ctx, &ctx->idx_ws, layout_index(nullptr, nbc, B, U, I, false, 16).bytes))
        return rc;
    const IndexWs w = layout_index(ctx->idx_ws.ptr, nbc, B, U, I, false, 16, 0, 0, lazy_spw(d));
    CREC_HIP(ctx, hipMemsetAsync(ctx->idx_ws.ptr, 0xff, w.bytes, st)); // (a partial last batch leaves its tail as it is: the same on both sides)
    if (big) {
        const LazyIndexOut o{w.u_desc, w.i_desc, w.j_desc, w.i_tmp, w.j_tmp, w.slot_pu, w.order, w.nslots};
        if (int rc = crec_lazy_index_big(ctx, u, i, j, N, B, nbc, w.bits_u, w.bits_q, w.spw, o, st))
            return rc;
    } else {
        const int cap = (2 * B + 63) / 64 * 64;
        const size_t lds = index_lds_bytes(cap);
        if (lds > ctx->index_lds_set) {
            CREC_HIP(ctx, hipFuncSetAttribute((const void *)bpr_index_batches, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ctx->index_lds_set = lds;
        }
        hipLaunchKernelGGL(bpr_index_batches, dim3((unsigned)(2 * nbc)), dim3(PREP_THREADS), lds, st, u, i, j, N, B, cap, w, 1);
    }
    const size_t n = (size_t)nbc * B;
    CREC_HIP(ctx, hipMemcpyAsync(u_desc, w.u_desc, sizeof(int4) * n, hipMemcpyDeviceToDevice, st));
    CREC_HIP(ctx, hipMemcpyAsync(i_desc, w.i_desc, sizeof(int4) * n, hipMemcpyDeviceToDevice, st));
    CREC_HIP(ctx, hipMemcpyAsync(j_desc, w.j_desc, sizeof(int4) * n, hipMemcpyDeviceToDevice, st));
    CREC_HIP(ctx, hipMemcpyAsync(slot_pu, w.slot_pu, sizeof(int2) * n, hipMemcpyDeviceToDevice, st));
    CREC_HIP(ctx, hipMemcpyAsync(nslots, w.nslots, sizeof(int32_t) * 2 * (size_t)nbc, hipMemcpyDeviceToDevice, st));
    CREC_HIP(ctx, hipStreamSynchronize(st));
    CREC_HIP(ctx, hipGetLastError());
    return CREC_OK;
}

} // extern "C"
