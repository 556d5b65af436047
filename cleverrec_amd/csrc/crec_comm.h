// Exchange layer of the row-sharded path (internal; see include/crec.h "row-sharded training").
//
// The sharded step has two real exchanges per batch -- table rows to the ranks that reference them, gradient rows
// back to their owners -- and per chunk of batches an exchange of the row-id lists.  All are all-to-all-v shaped:
// every rank sends a different, differently sized segment to every other rank.  On a node of MI355X that is the
// traffic pattern xGMI is good at (a full mesh: every pair of GPUs has its own link, so the G-1 segments of a rank
// leave on G-1 links at once), unlike a ring all-reduce of a dense table, which is bound by one link.
//
// Three transports behind one interface:
//   RCCL   one process per GPU (the production layout): grouped ncclSend / ncclRecv on the launch stream.
//   peer   one process per GPU (or the ctxs of one process): every rank maps every other rank's window (HIP IPC) and ONE kernel
//          per exchange publishes, waits on flags and reads the peers' segments over xGMI (crec_peer.hip); chosen with
//          crec_comm_peer_create / _connect -- microseconds per exchange where a grouped send / recv costs tens.
//   local  the ranks are ctxs of ONE process, each driven by its own host thread (one per GPU of a node, or -- in
//          the tests -- several ranks on one GPU): device-to-device copies between the ranks' buffers, host barriers.
// A ctx without a communicator is a world of one: the exchange is a device copy on the stream.
#pragma once
#include "crec_internal.h"
#include "crec_peer_dev.h"

// what TF1's dense Adam needs besides the variable, its slots and the gradient (lr_t = lr sqrt(1 - b2^t) / (1 - b1^t))
struct crec_adam_args {
    float lr_t, beta1, beta2, eps;
};

struct crec_xch {
    int rank = 0, world = 1;
    virtual ~crec_xch() {}
    // send[sdsp[p] .. sdsp[p] + scnt[p]) -> rank p; recv[rdsp[p] ..) <- rank p.  Counts and displacements in
    // elements of elem_bytes (4); buffers [dev]; enqueued on st (the local transport synchronises st).
    virtual int alltoallv(crec_ctx *ctx, const void *send, const int64_t *scnt, const int64_t *sdsp, void *recv,
                          const int64_t *rcnt, const int64_t *rdsp, int elem_bytes, hipStream_t st) = 0;
    virtual int allreduce_sum_f32(crec_ctx *ctx, float *buf, size_t n, hipStream_t st) = 0;
    // the largest payload a rank may hand to one exchange (the peer transport stages it in half of its window)
    virtual int64_t max_bytes() const { return INT64_MAX; }
    // Peer transport only (the others: false / nullptr / CREC_EUNSUPPORTED) -- for kernels that take part in an exchange themselves:
    //   begin      the view of the NEXT exchange; consumes its sequence number: the caller launches exactly one kernel that publishes
    //              this rank's segments (peer_publish) and waits for the peers' (peer_wait_all), crec_peer_dev.h
    //   next_half  where a producer kernel may leave the segments of the next exchange (byte offsets = what it will publish)
    //   alltoallv_inplace  all-to-all-v whose segments a previous kernel left in next_half() at their send displacements
    virtual bool begin(crec_peer_view *) { return false; }
    virtual char *next_half() { return nullptr; }
    virtual int alltoallv_inplace(crec_ctx *ctx, const int64_t *, const int64_t *, void *, const int64_t *, const int64_t *, hipStream_t)
    {
        return crec_fail(ctx, CREC_EUNSUPPORTED, "this transport has no window to exchange in place");
    }
    //   allreduce_adam  the replicated-Q step's exchange fused with its update: this rank's operand (n floats, its dQ) lies in next_half();
    //              ONE launch publishes it, forms the rank-ordered sum of all ranks' operands (bit-identical on every rank) and applies
    //              TF1's Adam to this rank's replica (Q, m, v) -- no reduced gradient in memory, no separate update launch
    virtual bool has_allreduce_adam(size_t /*n*/) { return false; }
    virtual int allreduce_adam(crec_ctx *ctx, size_t, float *, float *, float *, const crec_adam_args &, hipStream_t)
    {
        return crec_fail(ctx, CREC_EUNSUPPORTED, "this transport has no fused all-reduce + Adam");
    }
    // errors a transport only learns about asynchronously (the peer transport's waits give up after a time-out); synchronises st
    virtual int status(crec_ctx *, hipStream_t) { return CREC_OK; }
    // the same without waiting: the error word is sent to the host BEHIND the work enqueued so far and the call reports what earlier
    // such checks have brought back -- a time-out shows up one check late (or at the next status), the stream is never drained
    virtual int status_deferred(crec_ctx *ctx, hipStream_t st) { return status(ctx, st); }
};

// the ctx's exchange object: its local group, else RCCL (crec_comm_init), else the world-of-one copy
crec_xch *crec_xch_of(crec_ctx *ctx);
void crec_xch_release(crec_ctx *ctx);
