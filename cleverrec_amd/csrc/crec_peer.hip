// Peer transport of the exchange layer (crec_comm.h): the ranks' payloads cross xGMI as plain loads issued by ONE kernel per
// exchange, through windows every rank maps from every other rank (HIP IPC, one process per GPU; or raw pointers when the
// ranks are ctxs of one process) -- no proxy thread, no staging through a library's FIFO, no host round trip.
//
// Why: the row-sharded step has two dependent exchanges per batch and the replicated-Q step one all-reduce per ~8 us of
// compute; a grouped ncclSend / ncclRecv costs tens of microseconds whatever it carries, a launch that waits on seven
// flags and then reads seven segments costs a launch plus one xGMI round trip.  RCCL stays the portable transport (and the
// one that carries the unique id / handles at start-up); this one is chosen with crec_comm_peer_*.
//
// Protocol (pull): every rank owns a WINDOW (device memory, two halves used alternately) and a block of SIGNAL words
// (uncached device memory, [2 parities][2 phases][PEER_MAX] x 64 bit).  Exchange number n (counted per communicator, the same on every rank
// because exchanges are collective and stream-ordered) uses half n & 1:
//   1. copy-in   my segments for all peers -> my window half, back to back (16-byte aligned starts)
//   2. publish   when the LAST workgroup of the launch has finished its part of the copy (device-scope counter; the window
//                stores are written through at system scope and waited for, see copy_bytes) it stores {n, segment offset} into signal [n & 1][my rank] of every peer
//   3. pull      every workgroup waits for signal [n & 1][p] == n of every source p in its OWN signal block (local, uncached),
//                then reads its share of p's window half at the published offset straight into the receive buffer
//   all-reduce   same publish; every rank then reads all G operands and adds them in rank order (bit-identical sums
//                on every rank, which the replicated-Q step relies on)
// A window half is rewritten at exchange n + 2.  Safe: a peer publishes n + 1 only after its launch n has ended (stream
// order) -- pull included --, and my launch n + 1 waits for every peer's n + 1 before my launch n + 2 can start.  Nobody
// waits in step 1-2, so the waits of step 3 always end; they give up after `timeout` (20 s by default: ranks are only
// loosely in step, a peer may be allocating or planning on its host) and the communicator then reports the error.
#include "crec_comm.h"
#include "crec_peer_dev.h"

#include <unistd.h>

namespace {

struct PeerSegs {
    int64_t s_off[PEER_MAX], s_bytes[PEER_MAX]; // my segment for p inside `send`
    int64_t w_off[PEER_MAX];                    // ... and where it lies inside my window half
    int64_t r_off[PEER_MAX], r_bytes[PEER_MAX]; // what p sends me, inside `recv`
};

// Window traffic, two forms.  Default: every store into a window and every load from one is a SYSTEM-scope access (sc0 sc1: written
// through to memory / served from memory, as the compiler's own system-scope atomics are -- here 16 bytes at a time), so publishing
// needs no L2 write-back and reading no L2 invalidate: a wave waits for its stores (s_waitcnt vmcnt(0)) before the counter that leads
// to the flag.  FENCES (crec_comm_peer_fences, bench: CREC_PEER_FENCES=1): plain accesses bracketed by system-scope release /
// acquire fences issued by ONE wave per workgroup behind a barrier -- the textbook form, kept selectable at run time; a fence by
// every wave costs a full L2 write-back each (measured: 250 us for a 5 MB exchange).
// dst / src: wave-uniform bases (< 2 GiB behind them); WIN_DST / WIN_SRC: which side is a window
template <bool WIN_DST, bool WIN_SRC>
__device__ __forceinline__ void copy_bytes(char *dst, const char *src, int64_t bytes, int64_t tid, int64_t nth, bool fences)
{
    if ((((uintptr_t)dst | (uintptr_t)src | (uintptr_t)bytes) & 15) == 0) {
        const int64_t n = bytes >> 4;
        for (int64_t k = tid; k < n; k += nth) {
            const peer_u4v v = WIN_SRC ? peer_ld16(src, k << 4, fences) : ((const peer_u4v *)src)[k];
            if (WIN_DST)
                peer_st16(dst, k << 4, v, fences);
            else
                ((peer_u4v *)dst)[k] = v;
        }
    } else { // 4-byte elements
        const __amdgpu_buffer_rsrc_t rd = peer_rsrc(dst), rs = peer_rsrc(src);
        const int64_t n = bytes >> 2;
        for (int64_t k = tid; k < n; k += nth) {
            unsigned v;
            if (WIN_SRC && !fences)
                v = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(k << 2), 0, PEER_AUX_SYS);
            else
                v = ((const unsigned *)src)[k];
            if (WIN_DST && !fences)
                __builtin_amdgcn_raw_buffer_store_b32(v, rd, (int)(k << 2), 0, PEER_AUX_SYS);
            else
                ((unsigned *)dst)[k] = v;
        }
    }
}

// mode bit 0: all-reduce of n floats (send = recv = the operand), else all-to-all-v; bit 2: the all-reduce in two shots; bit 3: the
// all-to-all-v's segments already lie in this rank's window half (crec_xch::next_half, at their send displacements): no copy-in,
// workgroup 0 publishes at once -- and since no workgroup of such a launch waits for another one of it, its grid is not bounded.
// Otherwise every workgroup takes its grid-stride share of the copy-in and of every source's segment, at most PEER_MAX_BLOCKS
// workgroups, so that all of them are resident at once (a workgroup that waits for a peer must not keep one of its own launch --
// whose copy-in the peers wait for -- off the machine).
__global__ __launch_bounds__(PEER_THREADS) void peer_exchange(crec_peer_view v, PeerSegs a, const char *__restrict__ send, char *__restrict__ recv,
                                                              int mode, int64_t n)
{
    __shared__ unsigned long long flag_s[PEER_MAX];
    __shared__ int last_s, bad_s;
    const bool reduce = mode & 1, fences = v.fences != 0;
    const int tid = threadIdx.x, rank = v.rank, world = v.world;
    const int64_t gtid = (int64_t)blockIdx.x * PEER_THREADS + tid, gsz = (int64_t)gridDim.x * PEER_THREADS;
    // 1. copy-in (unless a previous kernel left the segments in the window half), 2. publish (by the workgroup that finishes its
    // part of the copy last; at once where there was nothing to copy), 3. wait for every source's flag
    if (!(mode & 8)) {
        for (int p = 0; p < (reduce ? 1 : world); ++p)
            if (a.s_bytes[p] > 0)
                copy_bytes<true, false>(v.my_half + a.w_off[p], send + a.s_off[p], a.s_bytes[p], gtid, gsz, fences);
    }
    peer_publish(v, 0, !(mode & 8), &last_s, [&](int p) { return a.w_off[reduce ? 0 : p]; });
    if (!peer_wait_all(v, 0, flag_s, &bad_s))
        return;
    // 4. pull: the peers behind this rank first, this rank's own segment last
    if (!reduce) {
        for (int s = 1; s <= world; ++s) {
            const int p = (rank + s) % world;
            if (a.r_bytes[p] > 0)
                copy_bytes<false, true>(recv + a.r_off[p], peer_seg(v, p, flag_s[p]), a.r_bytes[p], gtid, gsz, fences);
        }
    } else if (mode & 4) {
        // two-shot: this rank adds slice `rank` of all operands (rank order) into its window's result area and into `recv`, publishes
        // it (phase-1 flag), then collects the other slices from their owners: 2 (G-1)/G n floats cross the links instead of (G-1) n
        const int64_t n4 = n >> 2, per = (n4 + world - 1) / world;
        const int64_t roff = ((n * 4 + 15) >> 4) << 4; // result area, behind the operand
        float4 *out = (float4 *)recv;
        {
            const int64_t k0 = (int64_t)rank * per, k1 = (k0 + per) < n4 ? (k0 + per) : n4;
            for (int64_t k = k0 + gtid; k < k1; k += gsz) {
                peer_u4v x[PEER_MAX];
#pragma unroll
                for (int p = 0; p < PEER_MAX; ++p)
                    if (p < world)
                        x[p] = peer_ld16(peer_seg(v, p, flag_s[p]), k << 4, fences);
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int p = 0; p < PEER_MAX; ++p)
                    if (p < world)
                        acc = make_float4(acc.x + __uint_as_float(x[p].x), acc.y + __uint_as_float(x[p].y), acc.z + __uint_as_float(x[p].z),
                                          acc.w + __uint_as_float(x[p].w));
                out[k] = acc;
                const peer_u4v w = {__float_as_uint(acc.x), __float_as_uint(acc.y), __float_as_uint(acc.z), __float_as_uint(acc.w)};
                peer_st16(v.my_half + roff, (k - k0) << 4, w, fences);
            }
        }
        peer_publish(v, 1, true, &last_s, [&](int) { return roff; });
        if (!peer_wait_all(v, 1, flag_s, &bad_s))
            return;
        for (int s = 1; s < world; ++s) {
            const int p = (rank + s) % world;
            const int64_t k0 = (int64_t)p * per, k1 = (k0 + per) < n4 ? (k0 + per) : n4;
            if (k0 < k1)
                copy_bytes<false, true>((char *)(out + k0), peer_seg(v, p, flag_s[p]), (k1 - k0) << 4, gtid, gsz, fences);
        }
    } else if ((n & 3) == 0 && ((uintptr_t)recv & 15) == 0) {
        float4 *out = (float4 *)recv;
        for (int64_t k = gtid; k < (n >> 2); k += gsz) {
            peer_u4v x[PEER_MAX];
#pragma unroll
            for (int p = 0; p < PEER_MAX; ++p)
                if (p < world)
                    x[p] = peer_ld16(peer_seg(v, p, flag_s[p]), k << 4, fences);
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int p = 0; p < PEER_MAX; ++p) // rank order: every rank forms the same sum
                if (p < world)
                    acc = make_float4(acc.x + __uint_as_float(x[p].x), acc.y + __uint_as_float(x[p].y), acc.z + __uint_as_float(x[p].z),
                                      acc.w + __uint_as_float(x[p].w));
            out[k] = acc;
        }
    } else {
        float *out = (float *)recv;
        for (int64_t k = gtid; k < n; k += gsz) {
            float acc = 0.f;
            for (int p = 0; p < world; ++p) {
                const char *src = peer_seg(v, p, flag_s[p]);
                acc = acc + __uint_as_float(fences ? ((const unsigned *)src)[k] : __builtin_amdgcn_raw_buffer_load_b32(peer_rsrc(src), (int)(k << 2), 0, PEER_AUX_SYS));
            }
            out[k] = acc;
        }
    }
}

// ---- the replicated-Q step: all-reduce of the ranks' dQ fused with TF1's Adam on this rank's replica -------------------------------------
// The operand (n4 float4s) was left in this rank's window half by the step kernel; workgroup 0 publishes at once.  One shot: every
// rank adds all operands in rank order and updates all of Q.  Two shots (from three ranks on): this rank adds slice `rank`, updates it
// and leaves the sum in its window's result area; the other slices' sums are read from their owners and applied here -- every
// replica applies the same sums in the same arithmetic: bit-identical replicas, as with the separate all-reduce + bpr_adam_dense.
// The arithmetic is bpr_train.hip's adam4 (touched form), compiled like it with -ffp-contract=off.
__device__ __forceinline__ float peer_adam_quot(float num, float v, float eps)
{
#ifdef CREC_EXACT_MATH
    return num / (sqrtf(v) + eps);
#else
    return num * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(v) + eps);
#endif
}
struct PeerRow4 {
    float4 var, m, v;
};
__device__ __forceinline__ PeerRow4 peer_row_load(const float4 *Q, const float4 *m, const float4 *vv, int64_t k) { return PeerRow4{Q[k], m[k], vv[k]}; }
__device__ __forceinline__ void peer_adam4(float4 *Q, float4 *m, float4 *vv, int64_t k, PeerRow4 r, float4 g, const crec_adam_args &h)
{
    const float omb1 = 1.0f - h.beta1, omb2 = 1.0f - h.beta2;
    float *pv = &r.var.x, *pm = &r.m.x, *pw = &r.v.x;
    const float *pg = &g.x;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float mk = pm[c] * h.beta1;
        float vk = pw[c] * h.beta2;
        mk = mk + pg[c] * omb1;
        vk = vk + (pg[c] * pg[c]) * omb2;
        pm[c] = mk;
        pw[c] = vk;
        pv[c] = pv[c] - peer_adam_quot(h.lr_t * mk, vk, h.eps);
    }
    Q[k] = r.var;
    m[k] = r.m;
    vv[k] = r.v;
}
__device__ __forceinline__ float4 peer_sum_all(const crec_peer_view &v, const unsigned long long *flag_s, int64_t k, bool fences)
{
    peer_u4v x[PEER_MAX];
#pragma unroll
    for (int p = 0; p < PEER_MAX; ++p)
        if (p < v.world)
            x[p] = peer_ld16(peer_seg(v, p, flag_s[p]), k << 4, fences);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < PEER_MAX; ++p) // rank order: every rank forms the same sum
        if (p < v.world)
            acc = make_float4(acc.x + __uint_as_float(x[p].x), acc.y + __uint_as_float(x[p].y), acc.z + __uint_as_float(x[p].z),
                              acc.w + __uint_as_float(x[p].w));
    return acc;
}
// (a thread's first row of Q, m, v is requested BEFORE the wait for the peers' flags: it does not depend on them, and at ml-1m
// shape -- one or two float4s per thread -- that is most of the table traffic)
__global__ __launch_bounds__(PEER_THREADS) void peer_allreduce_adam(crec_peer_view v, int64_t n4, float4 *__restrict__ Q, float4 *__restrict__ m,
                                                                    float4 *__restrict__ vv, crec_adam_args h, int two_shot)
{
    __shared__ unsigned long long flag_s[PEER_MAX];
    __shared__ int last_s, bad_s;
    const bool fences = v.fences != 0;
    const int64_t gtid = (int64_t)blockIdx.x * PEER_THREADS + threadIdx.x, gsz = (int64_t)gridDim.x * PEER_THREADS;
    const int64_t per = two_shot ? (n4 + v.world - 1) / v.world : n4;
    const int64_t m0 = two_shot ? (int64_t)v.rank * per : 0, m1 = (m0 + per) < n4 ? (m0 + per) : n4; // the rows this rank sums itself
    PeerRow4 first{};
    if (m0 + gtid < m1)
        first = peer_row_load(Q, m, vv, m0 + gtid);
    peer_publish(v, 0, false, &last_s, [&](int) { return (int64_t)0; });
    if (!peer_wait_all(v, 0, flag_s, &bad_s))
        return;
    const int64_t roff = n4 << 4; // two shots: the result area, behind the operand
    for (int64_t k = m0 + gtid; k < m1; k += gsz) {
        const float4 acc = peer_sum_all(v, flag_s, k, fences);
        if (two_shot) {
            const peer_u4v w = {__float_as_uint(acc.x), __float_as_uint(acc.y), __float_as_uint(acc.z), __float_as_uint(acc.w)};
            peer_st16(v.my_half + roff, (k - m0) << 4, w, fences);
        }
        peer_adam4(Q, m, vv, k, k == m0 + gtid ? first : peer_row_load(Q, m, vv, k), acc, h);
    }
    if (!two_shot)
        return;
    peer_publish(v, 1, true, &last_s, [&](int) { return roff; });
    if (!peer_wait_all(v, 1, flag_s, &bad_s))
        return;
    // the other ranks' slices, all at once: slice lengths padded to whole waves, so that a wave reads from ONE peer (the window base
    // of a 16-byte access is wave-uniform) and every thread of the launch has a row of some slice
    const int64_t per_w = (per + 63) & ~(int64_t)63, tot = (int64_t)(v.world - 1) * per_w;
    for (int64_t t = gtid; t < tot; t += gsz) {
        const int sl = __builtin_amdgcn_readfirstlane((int)(t / per_w));
        const int p = (v.rank + 1 + sl) % v.world;
        const int64_t off = t - (int64_t)sl * per_w, k = (int64_t)p * per + off;
        if (off < per && k < n4) {
            const PeerRow4 r = peer_row_load(Q, m, vv, k);
            const peer_u4v x = peer_ld16(peer_seg(v, p, flag_s[p]), off << 4, fences);
            peer_adam4(Q, m, vv, k, r, make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w)), h);
        }
    }
}

struct PeerState { // what a rank owns; shared by the ranks of an in-process group through raw pointers
    char *win = nullptr;
    unsigned long long *sig = nullptr;
    int64_t win_bytes = 0;
    int device = -1;
};

struct PeerXch : crec_xch {
    PeerState own;
    crec_peer_view base{}; // win / sig of every rank as mapped here
    bool opened[PEER_MAX] = {false}; // mapped with hipIpcOpenMemHandle (to be closed)
    void *opened_win[PEER_MAX] = {nullptr}, *opened_sig[PEER_MAX] = {nullptr};
    int32_t *ctr = nullptr, *err = nullptr; // device: workgroup counter, error word
    int32_t *err_host = nullptr;            // page-locked landing place of the error word (a copy into pageable memory is not
                                            // stream-ordered alone: the runtime may drain the device's other streams for it -- with the
                                            // ranks as threads of one process, a peer's kernel that is waiting for THIS rank's next launch)
    uint32_t seq = 0;
    unsigned long long timeout_ticks = 20ull * 100000000ull; // 100 MHz
    bool connected = false;
    bool fences = false; // the textbook form of the window traffic (see copy_bytes)
    bool two_shot = true; // large all-reduces as reduce-scatter + gather (crec_comm_peer_fences bit 1 clears it: tests compare the two)

    ~PeerXch() override
    {
        for (int p = 0; p < PEER_MAX; ++p) {
            if (opened_win[p])
                (void)hipIpcCloseMemHandle(opened_win[p]);
            if (opened_sig[p])
                (void)hipIpcCloseMemHandle(opened_sig[p]);
        }
        if (own.win)
            (void)hipFree(own.win);
        if (own.sig)
            (void)hipFree(own.sig);
        if (ctr)
            (void)hipFree(ctr);
        if (err_host)
            (void)hipHostFree(err_host);
    }
    int64_t max_bytes() const override { return own.win_bytes / 2; }
    int status(crec_ctx *ctx, hipStream_t st) override
    {
        CREC_HIP(ctx, hipMemcpyAsync(err_host, err, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        CREC_HIP(ctx, hipStreamSynchronize(st));
        const int32_t e = *err_host;
        if (e)
            return crec_fail(ctx, CREC_EHIP, "peer transport: rank %d never published an exchange this rank waited for (gave up after %.0f s)",
                             e - 1, (double)timeout_ticks / 1e8);
        return CREC_OK;
    }
    int status_deferred(crec_ctx *ctx, hipStream_t st) override
    {
        const int32_t e = *(volatile int32_t *)err_host; // what earlier checks have brought back
        if (e)
            return crec_fail(ctx, CREC_EHIP, "peer transport: rank %d never published an exchange this rank waited for (gave up after %.0f s)",
                             e - 1, (double)timeout_ticks / 1e8);
        CREC_HIP(ctx, hipMemcpyAsync(err_host, err, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        return CREC_OK;
    }
    // the view of the next exchange; consumes its sequence number
    bool begin(crec_peer_view *v) override
    {
        if (!connected)
            return false;
        ++seq;
        if (seq == 0) // 0 is "no exchange yet".  The wrap keeps the parity alternating (0xffffffff is odd, so the next one is 2, not 1):
            seq = 2;  // two consecutive exchanges never share a window half and a pair of signal slots
        *v = base;
        v->half_off = (int64_t)(seq & 1u) * (own.win_bytes / 2);
        v->my_half = own.win + v->half_off;
        v->seq = seq;
        v->rank = rank;
        v->world = world;
        v->fences = fences ? 1 : 0;
        v->timeout_ticks = timeout_ticks;
        v->ctr = ctr;
        v->err = err;
        return true;
    }
    // where a producer kernel may leave the segments of the NEXT exchange (which then skips its copy-in)
    char *next_half() override { return connected ? own.win + (int64_t)((seq + 1 == 0 ? 2u : seq + 1) & 1u) * (own.win_bytes / 2) : nullptr; }
    int launch(crec_ctx *ctx, PeerSegs &a, const void *send, void *recv, int mode, int64_t n, int64_t work_bytes, hipStream_t st)
    {
        CREC_REQUIRE(ctx, connected, "peer transport: crec_comm_peer_connect has not completed on this ctx");
        crec_peer_view v;
        begin(&v);
        int nb = (int)((work_bytes + 16383) / 16384); // 16 KiB of the larger side per workgroup
        const int cap = (mode & 8) ? 8 * PEER_MAX_BLOCKS : PEER_MAX_BLOCKS;
        nb = nb < 1 ? 1 : (nb > cap ? cap : nb);
        hipLaunchKernelGGL(peer_exchange, dim3((unsigned)nb), dim3(PEER_THREADS), 0, st, v, a, (const char *)send, (char *)recv, mode, n);
        CREC_HIP(ctx, hipGetLastError());
        return CREC_OK;
    }
    int segs(crec_ctx *ctx, PeerSegs &a, const int64_t *scnt, const int64_t *sdsp, const int64_t *rcnt, const int64_t *rdsp, int64_t *work)
    {
        CREC_REQUIRE(ctx, scnt[rank] == rcnt[rank], "exchange: a rank's segment to itself has two sizes");
        int64_t at = 0, rtot = 0;
        for (int p = 0; p < world; ++p) {
            CREC_REQUIRE(ctx, scnt[p] >= 0 && rcnt[p] >= 0, "exchange: negative count");
            a.s_off[p] = sdsp[p] * 4;
            a.s_bytes[p] = scnt[p] * 4;
            a.w_off[p] = at;
            at = (at + scnt[p] * 4 + 15) / 16 * 16;
            a.r_off[p] = rdsp[p] * 4;
            a.r_bytes[p] = rcnt[p] * 4;
            rtot += rcnt[p] * 4;
        }
        CREC_REQUIRE(ctx, at <= own.win_bytes / 2, "peer transport: an exchange of %lld bytes does not fit half of the %lld-byte window",
                     (long long)at, (long long)own.win_bytes);
        *work = rtot > at ? rtot : at;
        return CREC_OK;
    }
    int alltoallv(crec_ctx *ctx, const void *send, const int64_t *scnt, const int64_t *sdsp, void *recv, const int64_t *rcnt,
                  const int64_t *rdsp, int elem_bytes, hipStream_t st) override
    {
        CREC_REQUIRE(ctx, elem_bytes == 4, "exchange: 4-byte elements");
        PeerSegs a{};
        int64_t work = 0;
        if (int rc = segs(ctx, a, scnt, sdsp, rcnt, rdsp, &work))
            return rc;
        return launch(ctx, a, send, recv, 0, 0, work, st);
    }
    // the segments already lie in next_half() at their send displacements (16-byte aligned), written by a previous kernel on st
    int alltoallv_inplace(crec_ctx *ctx, const int64_t *scnt, const int64_t *sdsp, void *recv, const int64_t *rcnt, const int64_t *rdsp,
                          hipStream_t st) override
    {
        PeerSegs a{};
        int64_t work = 0;
        if (int rc = segs(ctx, a, scnt, sdsp, rcnt, rdsp, &work))
            return rc;
        for (int p = 0; p < world; ++p) {
            CREC_REQUIRE(ctx, (a.s_off[p] & 15) == 0 && a.s_off[p] + a.s_bytes[p] <= own.win_bytes / 2, "in-place exchange: 16-byte aligned segments inside the window half");
            a.w_off[p] = a.s_off[p];
        }
        return launch(ctx, a, nullptr, recv, 8, 0, work, st);
    }
    // (two shots from three ranks and 64 KiB on, as allreduce_sum_f32; the operand and the result area must fit the window half)
    bool adam_two_shot(size_t n) const { return two_shot && world >= 3 && (int64_t)n * 4 >= (64 << 10); }
    bool has_allreduce_adam(size_t n) override
    {
        const int64_t per16 = (((int64_t)n / 4 + world - 1) / world) * 16;
        return connected && n > 0 && (n & 3) == 0 && (int64_t)n * 4 + (adam_two_shot(n) ? per16 : 0) <= own.win_bytes / 2;
    }
    int allreduce_adam(crec_ctx *ctx, size_t n, float *Q, float *m, float *v, const crec_adam_args &h, hipStream_t st) override
    {
        CREC_REQUIRE(ctx, has_allreduce_adam(n), "peer transport: fused all-reduce + Adam of %zu floats does not fit this communicator", n);
        CREC_REQUIRE(ctx, (((uintptr_t)Q | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "peer transport: 16-byte aligned tables");
        crec_peer_view pv;
        begin(&pv);
        const bool two = adam_two_shot(n);
        // a float4 of the table per thread; two shots: every workgroup waits for the peers' slices, which wait for ALL of this launch's
        // workgroups -- so they must be resident together (PEER_MAX_BLOCKS); one shot: no workgroup waits for one of its own launch
        int64_t nb = ((int64_t)n / 4 + PEER_THREADS - 1) / PEER_THREADS;
        nb = nb < 1 ? 1 : (nb > (two ? PEER_MAX_BLOCKS : 1024) ? (two ? PEER_MAX_BLOCKS : 1024) : nb);
        CREC_KLAUNCH(ctx, peer_allreduce_adam, dim3((unsigned)nb), dim3(PEER_THREADS), 0, st, pv, (int64_t)(n / 4), (float4 *)Q, (float4 *)m, (float4 *)v, h,
                     two ? 1 : 0);
        CREC_HIP(ctx, hipGetLastError());
        return CREC_OK;
    }
    int allreduce_sum_f32(crec_ctx *ctx, float *buf, size_t n, hipStream_t st) override
    {
        if (n == 0)
            return CREC_OK;
        CREC_REQUIRE(ctx, (int64_t)n * 4 <= own.win_bytes / 2, "peer transport: an all-reduce of %zu floats does not fit half of the window", n);
        PeerSegs a{};
        a.s_bytes[0] = (int64_t)n * 4;
        // from three ranks and 64 KiB on, in two shots (every rank adds one slice and the slices are gathered): same sums, same order
        const int64_t per16 = (((int64_t)n / 4 + world - 1) / world) * 16;
        const bool two = two_shot && world >= 3 && (n & 3) == 0 && ((uintptr_t)buf & 15) == 0 && (int64_t)n * 4 >= (64 << 10) &&
                         (((int64_t)n * 4 + 15) / 16 * 16 + per16) <= own.win_bytes / 2;
        return launch(ctx, a, buf, buf, two ? 5 : 1, (int64_t)n, two ? (int64_t)n * 4 * 2 : (int64_t)n * 4 * world, st);
    }
};

struct PeerBlob { // what a rank tells the others (CREC_PEER_BLOB_BYTES)
    hipIpcMemHandle_t win, sig;
    int64_t win_bytes;
    int32_t device, pid;
    char pad[CREC_PEER_BLOB_BYTES - 2 * sizeof(hipIpcMemHandle_t) - 16];
};
static_assert(sizeof(PeerBlob) == CREC_PEER_BLOB_BYTES, "blob size");

int peer_alloc(crec_ctx *ctx, int64_t window_bytes, PeerXch **out)
{
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, ctx->comm == nullptr && ctx->xch == nullptr, "communicator already initialised");
    CREC_REQUIRE(ctx, window_bytes >= (1 << 16) && window_bytes % 512 == 0 && window_bytes < ((int64_t)1 << 32),
                 "window_bytes: a multiple of 512, at least 64 KiB, below 4 GiB (a half is addressed with 32-bit offsets)");
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    PeerXch *x = new PeerXch();
    x->own.device = ctx->device;
    x->own.win_bytes = window_bytes;
    hipError_t e = hipMalloc((void **)&x->own.win, (size_t)window_bytes);
    if (e == hipSuccess)
        e = hipExtMallocWithFlags((void **)&x->own.sig, sizeof(unsigned long long) * 4 * PEER_MAX, hipDeviceMallocUncached);
    if (e == hipSuccess)
        e = hipMemset(x->own.sig, 0, sizeof(unsigned long long) * 4 * PEER_MAX);
    if (e == hipSuccess)
        e = hipHostMalloc((void **)&x->err_host, 64, hipHostMallocDefault);
    if (e == hipSuccess)
        e = hipMalloc((void **)&x->ctr, 4 * sizeof(int32_t));
    if (e == hipSuccess)
        e = hipMemset(x->ctr, 0, 4 * sizeof(int32_t));
    if (e == hipSuccess)
        e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        delete x;
        return crec_fail(ctx, e == hipErrorOutOfMemory ? CREC_ENOMEM : CREC_EHIP, "peer transport: allocating the window failed: %s", hipGetErrorString(e));
    }
    x->err = x->ctr + 1;
    *out = x;
    return CREC_OK;
}

} // namespace

extern "C" {

int crec_comm_peer_create(crec_ctx *ctx, int64_t window_bytes, void *blob_out)
{
    if (!ctx || !blob_out)
        return CREC_EINVAL;
    PeerXch *x = nullptr;
    if (int rc = peer_alloc(ctx, window_bytes, &x))
        return rc;
    PeerBlob b;
    std::memset(&b, 0, sizeof(b));
    hipError_t e = hipIpcGetMemHandle(&b.win, x->own.win);
    if (e == hipSuccess)
        e = hipIpcGetMemHandle(&b.sig, x->own.sig);
    if (e != hipSuccess) {
        delete x;
        return crec_fail(ctx, CREC_EHIP, "peer transport: hipIpcGetMemHandle failed: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", hipGetErrorString(e));
    }
    b.win_bytes = window_bytes;
    b.device = ctx->device;
    b.pid = (int32_t)getpid();
    std::memcpy(blob_out, &b, sizeof(b));
    ctx->xch = x; // not connected yet: any exchange before crec_comm_peer_connect is refused
    return CREC_OK;
}

int crec_comm_peer_connect(crec_ctx *ctx, int rank, int world, const void *blobs)
{
    if (!ctx || !blobs)
        return CREC_EINVAL;
    PeerXch *x = dynamic_cast<PeerXch *>(ctx->xch);
    CREC_REQUIRE(ctx, x != nullptr && !x->connected, "crec_comm_peer_create first (once)");
    CREC_REQUIRE(ctx, world >= 1 && world <= PEER_MAX && rank >= 0 && rank < world, "peer transport: world 1..%d", PEER_MAX);
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    const PeerBlob *b = (const PeerBlob *)blobs;
    for (int p = 0; p < world; ++p) {
        CREC_REQUIRE(ctx, b[p].win_bytes == x->own.win_bytes, "peer transport: rank %d has a window of %lld bytes, this rank %lld", p,
                     (long long)b[p].win_bytes, (long long)x->own.win_bytes);
        if (p == rank) {
            x->base.win[p] = x->own.win;
            x->base.sig[p] = x->own.sig;
            continue;
        }
        CREC_REQUIRE(ctx, b[p].pid != (int32_t)getpid(), "peer transport: rank %d lives in this process (use crec_comm_peer_connect_local)", p);
        void *w = nullptr, *s = nullptr;
        CREC_HIP(ctx, hipIpcOpenMemHandle(&w, b[p].win, hipIpcMemLazyEnablePeerAccess));
        x->opened_win[p] = w;
        CREC_HIP(ctx, hipIpcOpenMemHandle(&s, b[p].sig, hipIpcMemLazyEnablePeerAccess));
        x->opened_sig[p] = s;
        x->base.win[p] = (const char *)w;
        x->base.sig[p] = (unsigned long long *)s;
    }
    x->rank = rank;
    x->world = world;
    x->connected = true;
    ctx->rank = rank;
    ctx->world = world;
    return CREC_OK;
}

int crec_comm_peer_connect_local(crec_ctx **ctxs, int world, int64_t window_bytes)
{
    if (!ctxs || world < 1 || world > PEER_MAX)
        return CREC_EINVAL;
    std::vector<PeerXch *> xs((size_t)world, nullptr);
    for (int r = 0; r < world; ++r) {
        if (!ctxs[r])
            return CREC_EINVAL;
        if (int rc = peer_alloc(ctxs[r], window_bytes, &xs[(size_t)r])) {
            for (int q = 0; q < r; ++q)
                delete xs[(size_t)q];
            return rc;
        }
    }
    for (int r = 0; r < world; ++r) {
        for (int p = 0; p < world; ++p) {
            if (xs[(size_t)p]->own.device != xs[(size_t)r]->own.device) { // the ranks of one process on several GPUs of a node
                (void)hipSetDevice(xs[(size_t)r]->own.device);
                const hipError_t e = hipDeviceEnablePeerAccess(xs[(size_t)p]->own.device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
                    (void)hipGetLastError();
            }
            xs[(size_t)r]->base.win[p] = xs[(size_t)p]->own.win;
            xs[(size_t)r]->base.sig[p] = xs[(size_t)p]->own.sig;
        }
        xs[(size_t)r]->rank = r;
        xs[(size_t)r]->world = world;
        xs[(size_t)r]->connected = true;
        ctxs[r]->xch = xs[(size_t)r];
        ctxs[r]->rank = r;
        ctxs[r]->world = world;
    }
    (void)hipGetLastError();
    return CREC_OK;
}

int crec_comm_peer_fences(crec_ctx *ctx, int on)
{
    if (!ctx)
        return CREC_EINVAL;
    PeerXch *x = dynamic_cast<PeerXch *>(ctx->xch);
    CREC_REQUIRE(ctx, x != nullptr, "crec_comm_peer_fences: a peer-transport ctx");
    x->fences = (on & 1) != 0;
    x->two_shot = (on & 2) == 0;
    return CREC_OK;
}

int crec_comm_set_timeout(crec_ctx *ctx, double seconds)
{
    if (!ctx)
        return CREC_EINVAL;
    PeerXch *x = dynamic_cast<PeerXch *>(ctx->xch);
    CREC_REQUIRE(ctx, x != nullptr && seconds > 0 && seconds < 3600, "crec_comm_set_timeout: a peer-transport ctx and 0 < seconds < 3600");
    x->timeout_ticks = (unsigned long long)(seconds * 1e8);
    return CREC_OK;
}

// test / micro-benchmark hooks: one exchange of the ctx's communicator (any transport) on device buffers; counts and
// displacements in 4-byte elements, host arrays of `world` entries
int crec_debug_alltoallv(crec_ctx *ctx, const void *send, const int64_t *scnt, const int64_t *sdsp, void *recv, const int64_t *rcnt,
                         const int64_t *rdsp, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0 && scnt && sdsp && rcnt && rdsp, "null argument");
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    return crec_xch_of(ctx)->alltoallv(ctx, send, scnt, sdsp, recv, rcnt, rdsp, 4, (hipStream_t)stream);
}

int crec_debug_allreduce(crec_ctx *ctx, float *buf, int64_t n, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0 && buf && n >= 0, "null argument");
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    return crec_xch_of(ctx)->allreduce_sum_f32(ctx, buf, (size_t)n, (hipStream_t)stream);
}

int crec_comm_status(crec_ctx *ctx, void *stream)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    return crec_xch_of(ctx)->status(ctx, (hipStream_t)stream);
}

} // extern "C"
