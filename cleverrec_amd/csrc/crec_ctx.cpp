// Context, error reporting and workspace management of libcrec.so.
#include "crec_internal.h"

#include <cstdlib>
#include <dlfcn.h>
#include <mutex>

int crec_fail(crec_ctx *ctx, int code, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

int crec_reserve(crec_ctx *ctx, crec_devbuf *b, size_t bytes)
{
    if (bytes <= b->bytes)
        return CREC_OK;
    if (b->ptr) {
        // buffers may still be referenced by work in flight on a caller stream
        CREC_HIP(ctx, hipDeviceSynchronize());
        CREC_HIP(ctx, hipFree(b->ptr));
        b->ptr = nullptr;
        b->bytes = 0;
    }
    size_t want = bytes + bytes / 8;
    hipError_t e = hipMalloc(&b->ptr, want);
    if (e != hipSuccess) {
        b->ptr = nullptr;
        return crec_fail(ctx, CREC_ENOMEM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    }
    b->bytes = want;
    return CREC_OK;
}

int crec_side_stream(crec_ctx *ctx, hipStream_t *side)
{
    if (!ctx->idx_stream) {
        CREC_HIP(ctx, hipStreamCreateWithFlags(&ctx->idx_stream, hipStreamNonBlocking));
        for (hipEvent_t &e : ctx->idx_ev)
            CREC_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    *side = ctx->idx_stream;
    return CREC_OK;
}

static char g_create_err[256] = "no error";

namespace {
using push_fn = int (*)(const char *);
using pop_fn = int (*)();
push_fn g_push = nullptr;
pop_fn g_pop = nullptr;
std::once_flag g_roctx_once;
void load_roctx()
{
    const char *v = getenv("CREC_ROCTX");
    if (!v || !*v || *v == '0')
        return;
    for (const char *lib : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so"}) {
        if (void *h = dlopen(lib, RTLD_NOW | RTLD_LOCAL)) {
            g_push = (push_fn)dlsym(h, "roctxRangePushA");
            g_pop = (pop_fn)dlsym(h, "roctxRangePop");
            if (g_push && g_pop)
                return;
            g_push = nullptr, g_pop = nullptr;
        }
    }
}
} // namespace

crec_range::crec_range(const char *name) : on(false)
{
    std::call_once(g_roctx_once, load_roctx);
    if (g_push && g_pop) {
        g_push(name);
        on = true;
    }
}
crec_range::~crec_range()
{
    if (on)
        g_pop();
}

static hipEvent_t prof_event(crec_ctx *ctx)
{
    if (!ctx->prof.pool.empty()) {
        hipEvent_t e = ctx->prof.pool.back();
        ctx->prof.pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void crec_prof_begin(crec_ctx *ctx, int cls, hipStream_t st)
{
    if (!ctx->prof.on)
        return;
    crec_profile::rec r{cls, prof_event(ctx), prof_event(ctx), false, ctx->prof.skip_next, ctx->prof.next_units};
    ctx->prof.skip_next = false;
    ctx->prof.next_units = 1;
    ctx->prof.pending.push_back(r);
    ctx->prof.open = true;
    (void)st;
}

bool crec_prof_events(crec_ctx *ctx, hipEvent_t *a, hipEvent_t *b)
{
    if (!ctx || !ctx->prof.on || !ctx->prof.open || ctx->prof.pending.empty() || ctx->prof.pending.back().ext)
        return false;
    ctx->prof.pending.back().ext = true;
    *a = ctx->prof.pending.back().a;
    *b = ctx->prof.pending.back().b;
    return true;
}

void crec_prof_end(crec_ctx *ctx, hipStream_t st)
{
    if (!ctx->prof.on || !ctx->prof.open)
        return;
    ctx->prof.open = false;
    if (ctx->prof.pending.back().ext)
        return;
    // no launch claimed the events (a helper launched with plain hipLaunchKernelGGL): drop the record
    ctx->prof.pool.push_back(ctx->prof.pending.back().a);
    ctx->prof.pool.push_back(ctx->prof.pending.back().b);
    ctx->prof.pending.pop_back();
    (void)st;
}

// Per-kernel time = elapsed time between the two events attached to each dispatch (its begin / end timestamps).
int crec_prof_flush(crec_ctx *ctx, hipStream_t st)
{
    if (!ctx->prof.on || ctx->prof.pending.empty())
        return CREC_OK;
    CREC_HIP(ctx, hipStreamSynchronize(st));
    for (auto &r : ctx->prof.pending) {
        float ms = 0.f;
        if (r.ext && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && !r.skip) {
            ctx->prof.total_ms[r.cls] += ms > 0.f ? ms : 0.f;
            ctx->prof.launches[r.cls] += 1;
            ctx->prof.units[r.cls] += r.units;
        }
        ctx->prof.pool.push_back(r.a);
        ctx->prof.pool.push_back(r.b);
    }
    ctx->prof.pending.clear();
    ctx->prof.open = false;
    return CREC_OK;
}

extern "C" {

int crec_profile_enable(crec_ctx *ctx, int on)
{
    if (!ctx)
        return CREC_EINVAL;
    ctx->prof.on = on != 0;
    for (int k = 0; k < CREC_K_COUNT; ++k) {
        ctx->prof.total_ms[k] = 0;
        ctx->prof.launches[k] = 0;
        ctx->prof.units[k] = 0;
    }
    return CREC_OK;
}

int crec_profile_read(const crec_ctx *ctx, int kernel_class, double *total_ms, int64_t *launches)
{
    if (!ctx || kernel_class < 0 || kernel_class >= CREC_K_COUNT || !total_ms || !launches)
        return CREC_EINVAL;
    *total_ms = ctx->prof.total_ms[kernel_class];
    *launches = ctx->prof.launches[kernel_class];
    return CREC_OK;
}

int crec_profile_read_units(const crec_ctx *ctx, int kernel_class, int64_t *units)
{
    if (!ctx || kernel_class < 0 || kernel_class >= CREC_K_COUNT || !units)
        return CREC_EINVAL;
    *units = ctx->prof.units[kernel_class];
    return CREC_OK;
}

int crec_version(void) { return CREC_VERSION; }

int crec_ctx_create(int device, crec_ctx **out)
{
    if (!out)
        return CREC_EINVAL;
    *out = nullptr;
    crec_ctx *ctx = new (std::nothrow) crec_ctx();
    if (!ctx)
        return CREC_ENOMEM;
    ctx->device = device;
    crec_rng_seed(ctx, 5489u);
    const auto env_int = [](const char *name, int dflt) {
        const char *v = getenv(name);
        return v && *v ? atoi(v) : dflt;
    };
    ctx->knob_ch = env_int("CREC_CH", 0);
    ctx->knob_wide = env_int("CREC_WIDE", -1);
    ctx->knob_wide_round = env_int("CREC_WIDE_ROUND", 0);
    ctx->knob_eval_shape = env_int("CREC_EVAL_SHAPE", 0);
    ctx->knob_eval_no_bitmap = env_int("CREC_EVAL_NO_BITMAP", 0);
    ctx->knob_ncf_generic = env_int("CREC_NCF_GENERIC", 0);
    ctx->knob_shard_unfused = env_int("CREC_SHARD_UNFUSED", 0);
    ctx->knob_big_index = env_int("CREC_BIG_INDEX", 0);
    ctx->knob_dp_unfused = env_int("CREC_DP_UNFUSED", 0);
    ctx->knob_idx_one_wg = env_int("CREC_IDX_ONE_WG", 0);
    ctx->knob_idx_overlap = env_int("CREC_IDX_OVERLAP", 0);
    ctx->knob_no_zero_copy = env_int("CREC_NO_ZERO_COPY", 0);
    ctx->knob_dsampler_general = env_int("CREC_DSAMPLER_GENERAL", 0);
    ctx->knob_flow = env_int("CREC_FLOW", -1);
    ctx->knob_flow_steps = env_int("CREC_FLOW_STEPS", 0);
    ctx->knob_lazy_window = env_int("CREC_LAZY_WINDOW", -1);
    ctx->knob_flow_abl = env_int("CREC_FLOW_ABL", 0);
    if (device >= 0) {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || device >= n) {
            snprintf(g_create_err, sizeof(g_create_err), "hipGetDeviceCount -> %s, %d device(s), asked for %d",
                     hipGetErrorString(e), n, device);
            delete ctx;
            return CREC_ENODEVICE;
        }
        hipDeviceProp_t prop;
        if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
            snprintf(g_create_err, sizeof(g_create_err), "hipSetDevice/hipGetDeviceProperties(%d) -> %s", device,
                     hipGetErrorString(e));
            delete ctx;
            return CREC_ENODEVICE;
        }
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            // the code object is gfx950-only: say so now rather than at the first launch
            snprintf(g_create_err, sizeof(g_create_err), "device %d is %s; this library is built for gfx950 only", device,
                     prop.gcnArchName);
            delete ctx;
            return CREC_ENODEVICE;
        }
        ctx->num_cus = prop.multiProcessorCount;
    }
    *out = ctx;
    return CREC_OK;
}

void crec_ctx_destroy(crec_ctx *ctx)
{
    if (!ctx)
        return;
    if (ctx->device >= 0) {
        (void)hipSetDevice(ctx->device);
        (void)hipDeviceSynchronize();
        (void)crec_comm_destroy(ctx);
        crec_devbuf *bufs[] = {&ctx->idx_ws, &ctx->idx_ws2, &ctx->big_ws, &ctx->contrib, &ctx->loss_ws,
                               &ctx->misc,   &ctx->eval_ws, &ctx->ks_dev, &ctx->shadow, &ctx->dq, &ctx->shard_ws, &ctx->shard_ws2, &ctx->lr_hist_dev,
                               &ctx->shard_flag, &ctx->flow_hist, &ctx->flow_deal, &ctx->loss_parts, &ctx->lazy_done, &ctx->tf1_last, &ctx->lazy_wpart};
        for (crec_devbuf *b : bufs)
            if (b->ptr)
                (void)hipFree(b->ptr);
        for (auto &r : ctx->prof.pending) {
            (void)hipEventDestroy(r.a);
            (void)hipEventDestroy(r.b);
        }
        for (hipEvent_t e : ctx->prof.pool)
            (void)hipEventDestroy(e);
        for (hipEvent_t e : ctx->idx_ev)
            if (e)
                (void)hipEventDestroy(e);
        if (ctx->idx_stream)
            (void)hipStreamDestroy(ctx->idx_stream);
        if (ctx->pinned)
            (void)hipHostFree(ctx->pinned);
        if (ctx->pinned64)
            (void)hipHostFree(ctx->pinned64);
        if (ctx->flow_err)
            (void)hipHostFree(ctx->flow_err);
        if (ctx->idx_done)
            (void)hipFree(ctx->idx_done);
    }
    delete ctx;
}

int crec_ctx_set_option(crec_ctx *ctx, const char *name, int64_t value)
{
    if (!ctx || !name)
        return CREC_EINVAL;
    const std::string n(name);
    if (n == "ch")
        ctx->knob_ch = (int)value;
    else if (n == "wide")
        ctx->knob_wide = (int)value;
    else if (n == "wide_round")
        ctx->knob_wide_round = (int)value;
    else if (n == "eval_shape")
        ctx->knob_eval_shape = (int)value;
    else if (n == "eval_no_bitmap")
        ctx->knob_eval_no_bitmap = (int)value;
    else if (n == "ncf_generic")
        ctx->knob_ncf_generic = (int)value;
    else if (n == "no_zero_copy")
        ctx->knob_no_zero_copy = (int)value;
    else if (n == "shard_unfused")
        ctx->knob_shard_unfused = (int)value;
    else if (n == "big_index")
        ctx->knob_big_index = (int)value;
    else if (n == "dp_unfused")
        ctx->knob_dp_unfused = (int)value;
    else if (n == "idx_one_wg")
        ctx->knob_idx_one_wg = (int)value;
    else if (n == "idx_overlap")
        ctx->knob_idx_overlap = (int)value;
    else if (n == "flow")
        ctx->knob_flow = (int)value;
    else if (n == "flow_deal") {
        ctx->knob_flow_deal = (int)value;
        ctx->flow_deal_key[0] = 0; // (made again at the next call)
    } else if (n == "lazy_window")
        ctx->knob_lazy_window = (int)value;
    else if (n == "flow_abl")
        ctx->knob_flow_abl = (int)value;
    else if (n == "tf1_sweep")
        ctx->knob_tf1_sweep = (int)value;
    else if (n == "no_call_tail")
        ctx->knob_no_call_tail = (int)value;
    else if (n == "idx_no_ids16")
        ctx->knob_idx_no_ids16 = (int)value;
    else if (n == "flow_steps")
        ctx->knob_flow_steps = (int)value;
    else if (n == "dsampler_general")
        ctx->knob_dsampler_general = (int)value;
    else if (n == "dsampler_budget")
        ctx->knob_dsampler_budget = (int)value;
    else
        return crec_fail(ctx, CREC_EINVAL, "unknown option '%s'", name);
    return CREC_OK;
}

// ctx == NULL: why the last crec_ctx_create failed
const char *crec_last_error(const crec_ctx *ctx) { return ctx ? ctx->err : g_create_err; }

} // extern "C"
