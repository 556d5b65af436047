// Exchange layer of the row-sharded path (internal; see include/crec.h "row-sharded training").
//
// The sharded step has two real exchanges per batch -- table rows to the ranks that reference them, gradient rows
// back to their owners -- and per chunk of batches an exchange of the row-id lists.  All are all-to-all-v shaped:
// every rank sends a different, differently sized segment to every other rank.  On a node of MI355X that is the
// traffic pattern xGMI is good at (a full mesh: every pair of GPUs has its own link, so the G-1 segments of a rank
// leave on G-1 links at once), unlike a ring all-reduce of a dense table, which is bound by one link.
//
// Two transports behind one interface:
//   RCCL   one process per GPU (the production layout): grouped ncclSend / ncclRecv on the launch stream.
//   local  the ranks are ctxs of ONE process, each driven by its own host thread (one per GPU of a node, or -- in
//          the tests -- several ranks on one GPU): device-to-device copies between the ranks' buffers, host barriers.
// A ctx without a communicator is a world of one: the exchange is a device copy on the stream.
#pragma once
#include "crec_internal.h"

struct crec_xch {
    int rank = 0, world = 1;
    virtual ~crec_xch() {}
    // send[sdsp[p] .. sdsp[p] + scnt[p]) -> rank p; recv[rdsp[p] ..) <- rank p.  Counts and displacements in
    // elements of elem_bytes (4); buffers [dev]; enqueued on st (the local transport synchronises st).
    virtual int alltoallv(crec_ctx *ctx, const void *send, const int64_t *scnt, const int64_t *sdsp, void *recv,
                          const int64_t *rcnt, const int64_t *rdsp, int elem_bytes, hipStream_t st) = 0;
    virtual int allreduce_sum_f32(crec_ctx *ctx, float *buf, size_t n, hipStream_t st) = 0;
};

// the ctx's exchange object: its local group, else RCCL (crec_comm_init), else the world-of-one copy
crec_xch *crec_xch_of(crec_ctx *ctx);
void crec_xch_release(crec_ctx *ctx);
