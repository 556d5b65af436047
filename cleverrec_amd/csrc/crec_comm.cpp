// Communicators of libcrec.so: RCCL (one process per GPU) and the in-process local group (crec_comm.h).
#include "crec_comm.h"

#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>

namespace {

// ---- world of one ---------------------------------------------------------------------------------------------
struct SelfXch : crec_xch {
    int alltoallv(crec_ctx *ctx, const void *send, const int64_t *scnt, const int64_t *sdsp, void *recv,
                  const int64_t *rcnt, const int64_t *rdsp, int elem_bytes, hipStream_t st) override
    {
        CREC_REQUIRE(ctx, scnt[0] == rcnt[0], "exchange: send and receive counts of the only rank differ");
        if (scnt[0] > 0)
            CREC_HIP(ctx, hipMemcpyAsync((char *)recv + rdsp[0] * elem_bytes, (const char *)send + sdsp[0] * elem_bytes,
                                         (size_t)scnt[0] * elem_bytes, hipMemcpyDeviceToDevice, st));
        return CREC_OK;
    }
    int allreduce_sum_f32(crec_ctx *, float *, size_t, hipStream_t) override { return CREC_OK; }
};

// ---- RCCL -----------------------------------------------------------------------------------------------------
// all-to-all-v = one group of point-to-point transfers: on the xGMI mesh the G-1 segments travel on G-1 links at once
struct RcclXch : crec_xch {
    ncclComm_t comm;
    int alltoallv(crec_ctx *ctx, const void *send, const int64_t *scnt, const int64_t *sdsp, void *recv,
                  const int64_t *rcnt, const int64_t *rdsp, int elem_bytes, hipStream_t st) override
    {
        CREC_REQUIRE(ctx, elem_bytes == 4, "exchange: 4-byte elements");
        CREC_REQUIRE(ctx, scnt[rank] == rcnt[rank], "exchange: a rank's segment to itself has two sizes");
        if (scnt[rank] > 0) // the rank's own segment never leaves the device
            CREC_HIP(ctx, hipMemcpyAsync((char *)recv + rdsp[rank] * 4, (const char *)send + sdsp[rank] * 4,
                                         (size_t)scnt[rank] * 4, hipMemcpyDeviceToDevice, st));
        if (world == 1)
            return CREC_OK;
        ncclResult_t r = ncclGroupStart();
        for (int p = 0; p < world && r == ncclSuccess; ++p) {
            if (p == rank)
                continue;
            if (scnt[p] > 0)
                r = ncclSend((const char *)send + sdsp[p] * 4, (size_t)scnt[p], ncclInt32, p, comm, st);
            if (r == ncclSuccess && rcnt[p] > 0)
                r = ncclRecv((char *)recv + rdsp[p] * 4, (size_t)rcnt[p], ncclInt32, p, comm, st);
        }
        const ncclResult_t e = ncclGroupEnd();
        if (r != ncclSuccess || e != ncclSuccess)
            return crec_fail(ctx, CREC_EHIP, "RCCL all-to-all-v failed: %s", ncclGetErrorString(r != ncclSuccess ? r : e));
        return CREC_OK;
    }
    int allreduce_sum_f32(crec_ctx *ctx, float *buf, size_t n, hipStream_t st) override
    {
        if (world == 1 || n == 0)
            return CREC_OK;
        if (ncclAllReduce(buf, buf, n, ncclFloat, ncclSum, comm, st) != ncclSuccess)
            return crec_fail(ctx, CREC_EHIP, "ncclAllReduce failed");
        return CREC_OK;
    }
};

// ---- local group: the ranks are ctxs of this process, one host thread each ------------------------------------
struct LocalGroup {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t gen = 0;
    bool broken = false;
    struct Pub {
        void *recv = nullptr;
        const int64_t *rcnt = nullptr, *rdsp = nullptr;
        std::vector<float> host;
    };
    std::vector<Pub> pub;
    bool step_failed = false; // some rank's part of the exchange in progress failed: every rank reports it (read behind the closing barrier)
    // a rank that cannot take part any more (its own HIP call failed before a barrier): the others must not wait 120 s for it
    void give_up()
    {
        std::lock_guard<std::mutex> lk(mu);
        broken = true;
        cv.notify_all();
    }
    void mark_failed()
    {
        std::lock_guard<std::mutex> lk(mu);
        step_failed = true;
    }
    // false: a rank did not arrive within the time-out (it failed or was never called); the group stays broken
    bool barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        if (broken)
            return false;
        const uint64_t g = gen;
        if (++arrived == world) {
            arrived = 0;
            ++gen;
            cv.notify_all();
            return true;
        }
        if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return gen != g || broken; }) || broken) {
            broken = true;
            cv.notify_all();
            return false;
        }
        return true;
    }
};

struct LocalXch : crec_xch {
    std::shared_ptr<LocalGroup> grp;
    int alltoallv(crec_ctx *ctx, const void *send, const int64_t *scnt, const int64_t *sdsp, void *recv,
                  const int64_t *rcnt, const int64_t *rdsp, int elem_bytes, hipStream_t st) override
    {
        // everything this rank enqueued so far is done: its receive buffer is free, its send buffer is filled
        if (hipStreamSynchronize(st) != hipSuccess) {
            grp->give_up();
            return crec_fail(ctx, CREC_EHIP, "local group: hipStreamSynchronize failed before the exchange");
        }
        grp->pub[rank].recv = recv;
        grp->pub[rank].rcnt = rcnt;
        grp->pub[rank].rdsp = rdsp;
        if (!grp->barrier())
            return crec_fail(ctx, CREC_EHIP, "local group: a rank did not reach the exchange");
        bool ok = true;
        for (int p = 0; p < world; ++p) {
            const LocalGroup::Pub &q = grp->pub[p];
            ok = ok && q.rcnt[rank] == scnt[p];
            if (ok && scnt[p] > 0)
                ok = hipMemcpyAsync((char *)q.recv + q.rdsp[rank] * elem_bytes, (const char *)send + sdsp[p] * elem_bytes,
                                    (size_t)scnt[p] * elem_bytes, hipMemcpyDefault, st) == hipSuccess;
        }
        ok = ok && hipStreamSynchronize(st) == hipSuccess;
        if (!ok)
            grp->mark_failed(); // a count mismatch is seen by one side only: all ranks return the same verdict
        if (!grp->barrier() || grp->step_failed) // every rank's segments have landed everywhere
            return crec_fail(ctx, CREC_EHIP, "local group: exchange failed (count mismatch, copy error or a missing rank)");
        return CREC_OK;
    }
    int allreduce_sum_f32(crec_ctx *ctx, float *buf, size_t n, hipStream_t st) override
    {
        std::vector<float> &mine = grp->pub[rank].host;
        mine.resize(n);
        if (hipMemcpyAsync(mine.data(), buf, n * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
            grp->give_up();
            return crec_fail(ctx, CREC_EHIP, "local group: reading this rank's operand of the all-reduce failed");
        }
        if (!grp->barrier())
            return crec_fail(ctx, CREC_EHIP, "local group: a rank did not reach the all-reduce");
        std::vector<float> sum(n, 0.f);
        for (int p = 0; p < world; ++p) // rank order: every rank gets the bit-identical sum
            for (size_t k = 0; k < n; ++k)
                sum[k] = sum[k] + grp->pub[p].host[k];
        if (hipMemcpy(buf, sum.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            grp->mark_failed();
        if (!grp->barrier() || grp->step_failed) // nobody refills its host copy before everybody has read it
            return crec_fail(ctx, CREC_EHIP, "local group: all-reduce failed");
        return CREC_OK;
    }
};

} // namespace

crec_xch *crec_xch_of(crec_ctx *ctx)
{
    if (!ctx->xch) {
        if (ctx->comm) {
            RcclXch *x = new RcclXch();
            x->comm = (ncclComm_t)ctx->comm;
            x->rank = ctx->rank;
            x->world = ctx->world;
            ctx->xch = x;
        } else {
            ctx->xch = new SelfXch();
        }
    }
    return ctx->xch;
}

void crec_xch_release(crec_ctx *ctx)
{
    delete ctx->xch;
    ctx->xch = nullptr;
}

extern "C" {

int crec_comm_unique_id(crec_ctx *ctx, void *id_out)
{
    if (!ctx || !id_out)
        return CREC_EINVAL;
    static_assert(sizeof(ncclUniqueId) == CREC_UNIQUE_ID_BYTES, "unique id size");
    if (ncclGetUniqueId((ncclUniqueId *)id_out) != ncclSuccess)
        return crec_fail(ctx, CREC_EHIP, "ncclGetUniqueId failed");
    return CREC_OK;
}

int crec_comm_init(crec_ctx *ctx, int rank, int world, const void *id)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, ctx->device >= 0, "host-only ctx: no device bound");
    CREC_REQUIRE(ctx, id && world >= 1 && rank >= 0 && rank < world, "bad rank/world");
    CREC_REQUIRE(ctx, ctx->comm == nullptr && ctx->xch == nullptr, "communicator already initialised");
    CREC_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    ncclResult_t r = ncclCommInitRank(&comm, world, uid, rank);
    if (r != ncclSuccess)
        return crec_fail(ctx, CREC_EHIP, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->world = world;
    return CREC_OK;
}

int crec_comm_init_local(crec_ctx **ctxs, int world)
{
    if (!ctxs || world < 1)
        return CREC_EINVAL;
    for (int r = 0; r < world; ++r) {
        if (!ctxs[r])
            return CREC_EINVAL;
        CREC_REQUIRE(ctxs[r], ctxs[r]->device >= 0, "host-only ctx: no device bound");
        CREC_REQUIRE(ctxs[r], ctxs[r]->comm == nullptr && ctxs[r]->xch == nullptr, "communicator already initialised");
    }
    auto grp = std::make_shared<LocalGroup>();
    grp->world = world;
    grp->pub.resize((size_t)world);
    for (int r = 0; r < world; ++r) {
        LocalXch *x = new LocalXch();
        x->grp = grp;
        x->rank = r;
        x->world = world;
        ctxs[r]->xch = x;
        ctxs[r]->rank = r;
        ctxs[r]->world = world;
    }
    return CREC_OK;
}

int crec_comm_destroy(crec_ctx *ctx)
{
    if (!ctx)
        return CREC_EINVAL;
    crec_xch_release(ctx);
    if (ctx->comm) {
        ncclCommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->rank = 0;
    ctx->world = 1;
    return CREC_OK;
}

} // extern "C"
