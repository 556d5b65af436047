// Host side of the hot path: the NumPy-legacy MT19937 stream and the three epoch samplers,
// bit-exact with the reference on the same RNG state.
//
// The host form.  Since round 3 the model classes draw these streams on the device (mt_sampler.hip: same words, same
// columns, same final state); this file stays as the form that has no limits (any catalogue size), for SBPR's sampler, for
// `sampler_mode=mt19937_host`, and as the RNG the ctx exposes (crec_rng_*).  One RNG thread walks the rejection loops and the
// shuffle's draws; the swaps and the gather of the output columns are pipelined behind it on two more threads.
//
// Reference: utils/sampler.py:10-99 (samplers), numpy RandomState (randint / permutation).
#include <cstdlib>
#include <sched.h>
#include <sys/resource.h>
#include <memory>
#include <new>

#include "crec_internal.h"

#include <atomic>
#include <thread>
#include <chrono>
#include <vector>

namespace {

constexpr int MT_N = 624, MT_M = 397;

inline void mt_refill(crec_mt19937 &r)
{
    uint32_t *mt = r.key;
    auto twist = [](uint32_t a, uint32_t b, uint32_t c) {
        uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
        return c ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
    };
    int k = 0;
    for (; k < MT_N - MT_M; ++k)
        mt[k] = twist(mt[k], mt[k + 1], mt[k + MT_M]);
    for (; k < MT_N - 1; ++k)
        mt[k] = twist(mt[k], mt[k + 1], mt[k + MT_M - MT_N]);
    mt[MT_N - 1] = twist(mt[MT_N - 1], mt[0], mt[MT_M - 1]);
    r.pos = 0;
}

inline uint32_t mt_next(crec_mt19937 &r)
{
    if (__builtin_expect(r.pos >= MT_N, 0))
        mt_refill(r);
    uint32_t y = r.key[r.pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

inline uint32_t mask_of(uint32_t max)
{
    uint32_t m = max;
    m |= m >> 1;
    m |= m >> 2;
    m |= m >> 4;
    m |= m >> 8;
    m |= m >> 16;
    return m;
}

// masked rejection draw in [0, max]; consumes no word when max == 0
inline uint32_t draw(crec_mt19937 &r, uint32_t max, uint32_t mask)
{
    if (max == 0)
        return 0;
    uint32_t v;
    do {
        v = mt_next(r) & mask;
    } while (v > max);
    return v;
}

// legacy shuffle of arange(n): for k = n-1 .. 1: swap(a[k], a[draw in [0, k]]).
// The swap targets depend only on the RNG stream, never on the array, so they are drawn a
// block ahead and prefetched: the swaps themselves still run strictly in order, but on
// lines that are already on their way (the array is far larger than the host caches).
template <typename T> void permutation(crec_mt19937 &r, int64_t n, T *a, std::atomic<int64_t> *front = nullptr)
{
    for (int64_t k = 0; k < n; ++k)
        a[k] = (T)k;
    constexpr int BLK = 64;
    uint32_t jb[BLK];
    for (int64_t k = n - 1; k >= 1;) {
        const int m = (int)(k < BLK ? k : BLK);
        for (int t = 0; t < m; ++t) {
            uint32_t kk = (uint32_t)(k - t);
            jb[t] = draw(r, kk, mask_of(kk));
            __builtin_prefetch(&a[jb[t]], 1, 1);
        }
        for (int t = 0; t < m; ++t) {
            T *x = &a[k - t], *y = &a[jb[t]];
            T tmp = *x;
            *x = *y;
            *y = tmp;
        }
        k -= m;
        if (front) // positions above k are never touched again: they are final
            front->store(k + 1, std::memory_order_release);
    }
    if (front)
        front->store(0, std::memory_order_release);
}

// permutation(n) + "for k in [0, n): body(k, perm[k])" with the body spread over worker threads that trail the
// shuffle: the legacy shuffle walks k downwards and never touches a position above k again, so perm[k..n) is final
// as soon as step k is done and can be consumed while the (strictly serial) shuffle is still running.  The RNG
// stream and the result are exactly those of the serial code.
// n / d for every 32-bit n by one multiplication and two shifts (Granlund-Montgomery, the round-up form): the gather
// divides every shuffled index by neg_ratio, and a hardware 32-bit division costs 10-30 ns where this costs one.
struct FastDiv {
    uint32_t m, s1, s2;
    explicit FastDiv(uint32_t d)
    {
        d = d ? d : 1; // neg_ratio 0: an empty sample, nothing is divided
        uint32_t L = 0;
        while (((uint64_t)1 << L) < d)
            ++L;
        m = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << L) - d)) / d + 1);
        s1 = L < 1 ? L : 1;
        s2 = L - s1;
    }
    uint32_t div(uint32_t n) const
    {
        const uint32_t t = (uint32_t)(((uint64_t)m * n) >> 32);
        return (t + ((n - t) >> s1)) >> s2;
    }
};

// `ahead(sidx)`: prefetches of what body(k, sidx) will read -- the gather's reads are scattered over arrays far larger than
// the caches, and without them every element waits for its own misses one after the other.
struct NoAhead {
    void operator()(uint32_t) const {}
};
constexpr int GATHER_AHEAD = 24;
template <class Body, class Ahead = NoAhead>
void shuffled_apply(crec_mt19937 &r, int64_t n, uint32_t *perm, Body body, Ahead ahead = Ahead())
{
    static const int env_threads = getenv("CREC_SAMPLER_THREADS") ? atoi(getenv("CREC_SAMPLER_THREADS")) : -1;
    // one gather worker (+ the swap thread below) keeps up with the RNG thread on the GPU host: 45 -> 34 ms per
    // ml-1m epoch; more workers only add contention
    int workers = env_threads >= 0 ? env_threads : ((int)std::thread::hardware_concurrency() >= 4 ? 1 : 0);
    workers = workers > 8 ? 8 : workers;
    const auto serial_gather = [&] {
        for (int64_t k = 0; k < n; ++k) {
            if (k + GATHER_AHEAD < n)
                ahead(perm[k + GATHER_AHEAD]);
            body(k, perm[k]);
        }
    };
    if (workers < 1 || n < (1 << 16)) {
        permutation(r, n, perm);
        serial_gather();
        return;
    }
    constexpr int64_t CHUNK = 1 << 14;
    const int64_t n_chunks = (n + CHUNK - 1) / CHUNK;
    std::atomic<int64_t> front{n}, next{n_chunks - 1};
    // the shuffle itself in two stages: this thread owns the RNG and only draws the swap targets (they depend on the
    // stream alone), a second thread applies the swaps behind it -- generation and the cache-missing swaps overlap
    std::vector<uint32_t> target;
    std::atomic<int64_t> drawn{n}; // steps k >= drawn have their target
    std::thread swapper;
    try { // nothing may escape through the C ABI: without memory or threads, do it serially
        target.resize((size_t)n);
        swapper = std::thread([&] {
            for (int64_t k = 0; k < n; ++k)
                perm[k] = (uint32_t)k;
            constexpr int64_t AHEAD = 64;
            int64_t avail = n; // local copy of `drawn`
            for (int64_t k = n - 1; k >= 1; --k) {
                while (avail > k) {
                    avail = drawn.load(std::memory_order_acquire);
                    if (avail > k)
                        std::this_thread::yield();
                }
                if (k - AHEAD >= avail)
                    __builtin_prefetch(&perm[target[k - AHEAD]], 1, 1);
                const uint32_t j = target[k], tmp = perm[k];
                perm[k] = perm[j];
                perm[j] = tmp;
                if ((k & 63) == 0) // positions >= k are never touched again: they are final
                    front.store(k, std::memory_order_release);
            }
            front.store(0, std::memory_order_release);
        });
    } catch (...) {
        permutation(r, n, perm);
        serial_gather();
        return;
    }
    const auto gather = [&] { // chunks from the top down, each as soon as the shuffle has passed it
        for (;;) {
            const int64_t c = next.fetch_sub(1, std::memory_order_relaxed);
            if (c < 0)
                return;
            const int64_t lo = c * CHUNK, hi = (lo + CHUNK) < n ? (lo + CHUNK) : n;
            while (front.load(std::memory_order_acquire) > lo)
                std::this_thread::yield();
            for (int64_t k = lo; k < hi; ++k) {
                if (k + 2 * GATHER_AHEAD < hi)
                    __builtin_prefetch(&perm[k + 2 * GATHER_AHEAD], 0, 1);
                if (k + GATHER_AHEAD < hi)
                    ahead(perm[k + GATHER_AHEAD]);
                body(k, perm[k]);
            }
        }
    };
    std::vector<std::thread> pool;
    try {
        for (int w = 0; w < workers; ++w)
            pool.emplace_back(gather);
    } catch (...) { // fewer workers than asked for: this thread picks up what is left at the end
    }
    for (int64_t k = n - 1; k >= 1; --k) {
        const uint32_t kk = (uint32_t)k;
        target[k] = draw(r, kk, mask_of(kk));
        if ((k & 1023) == 0)
            drawn.store(k, std::memory_order_release);
    }
    drawn.store(1, std::memory_order_release);
    swapper.join();
    gather(); // whatever the workers have not claimed (everything, if there are none)
    for (auto &t : pool)
        t.join();
}

struct Csr {
    const int64_t *ptr;
    const int32_t *items;
    int64_t n_users;
    const int32_t *user_ids;
    int32_t item_nums;
};

// Draws the negatives of every positive in reference order.  neg[e*neg_ratio + t] is the
// t-th negative of positive e.  Returns false if some user cannot supply neg_ratio negatives.
bool draw_negatives(crec_mt19937 &r, const Csr &c, int neg_ratio, int32_t *neg, int32_t *stamp)
{
    const uint32_t max = (uint32_t)(c.item_nums - 1), mask = mask_of(max);
    int64_t out = 0;
    for (int64_t k = 0; k < c.n_users; ++k) {
        const int32_t tag = (int32_t)(k + 1);
        int64_t uniq = 0;
        for (int64_t e = c.ptr[k]; e < c.ptr[k + 1]; ++e) {
            uniq += (stamp[c.items[e]] != tag);
            stamp[c.items[e]] = tag; // seen_items = set(ui_train[u])
        }
        if (c.ptr[k + 1] > c.ptr[k] && uniq + neg_ratio > c.item_nums)
            return false;
        for (int64_t e = c.ptr[k]; e < c.ptr[k + 1]; ++e) {
            int32_t *grp = neg + out; // random_j of this positive
            for (int t = 0; t < neg_ratio; ++t) {
                for (;;) {
                    int32_t j = (int32_t)draw(r, max, mask);
                    bool bad = stamp[j] == tag;
                    for (int q = 0; q < t; ++q)
                        bad |= (grp[q] == j);
                    if (!bad) {
                        grp[t] = j;
                        break;
                    }
                }
            }
            out += neg_ratio;
        }
    }
    return true;
}

template <typename T> struct Free {
    void operator()(T *p) const { std::free(p); }
};
template <typename T> using buf = std::unique_ptr<T[], Free<T>>;
// (Keeping these arrays in the ctx from one epoch to the next was tried: no 40 MB of malloc and first-touch per call -- 120 -> 78 ms
// per epoch on the build container's Xeon, but prefetched epochs ~33 -> ~55 ms on the GPU host's EPYC in two interleaved A/B runs
// (tools/sampler_ab.sh; presumably last epoch's lines still sit in the caches of the cores that ran last epoch's threads and this
// epoch's writes pay for them, where freshly mapped pages have no sharers).  The host is noisy -- the same build gives 28 or 63 ms
// epochs minutes apart -- so only interleaved runs on one box mean anything.)
template <typename T> buf<T> alloc(int64_t n, bool zero = false)
{
    size_t bytes = sizeof(T) * (size_t)(n > 0 ? n : 1);
    return buf<T>((T *)(zero ? std::calloc(bytes, 1) : std::malloc(bytes)));
}

int check_csr(crec_ctx *ctx, const Csr &c, int neg_ratio, const void *a, const void *b, const void *d, int64_t *n_out)
{
    CREC_REQUIRE(ctx, c.ptr && c.user_ids && a && b && d && n_out, "null argument");
    CREC_REQUIRE(ctx, c.n_users >= 0 && c.item_nums >= 1 && neg_ratio >= 0, "bad sizes");
    CREC_REQUIRE(ctx, c.items || c.ptr[c.n_users] == 0, "null items");
    for (int64_t k = 0; k < c.n_users; ++k)
        CREC_REQUIRE(ctx, c.ptr[k + 1] >= c.ptr[k], "ptr not monotone at %lld", (long long)k);
    for (int64_t e = 0; e < c.ptr[c.n_users]; ++e)
        CREC_REQUIRE(ctx, c.items[e] >= 0 && c.items[e] < c.item_nums, "item id %d out of range", c.items[e]);
    return CREC_OK;
}

// edge -> user id, so the shuffled u column is one gather
buf<int32_t> edge_users(const Csr &c)
{
    auto eu = alloc<int32_t>(c.ptr[c.n_users]);
    if (eu)
        for (int64_t k = 0; k < c.n_users; ++k)
            for (int64_t e = c.ptr[k]; e < c.ptr[k + 1]; ++e)
                eu[e] = c.user_ids[k];
    return eu;
}

} // namespace

extern "C" {

// test hook: FastDiv(d).div(n) for n in ns[0 .. count) (tests/test_host.py checks it against the hardware division)
int crec_debug_fastdiv(uint32_t d, const uint32_t *ns, int64_t count, uint32_t *out)
{
    if (d == 0 || !ns || !out)
        return CREC_EINVAL;
    const FastDiv f(d);
    for (int64_t k = 0; k < count; ++k)
        out[k] = f.div(ns[k]);
    return CREC_OK;
}

int crec_rng_seed(crec_ctx *ctx, uint32_t seed)
{
    if (!ctx)
        return CREC_EINVAL;
    uint32_t *mt = ctx->rng.key;
    mt[0] = seed;
    for (int k = 1; k < MT_N; ++k)
        mt[k] = 1812433253u * (mt[k - 1] ^ (mt[k - 1] >> 30)) + (uint32_t)k;
    ctx->rng.pos = MT_N;
    return CREC_OK;
}

int crec_rng_set_state(crec_ctx *ctx, const uint32_t key[624], int32_t pos)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, key && pos >= 0 && pos <= MT_N, "bad MT19937 state");
    std::memcpy(ctx->rng.key, key, sizeof(ctx->rng.key));
    ctx->rng.pos = pos;
    return CREC_OK;
}

int crec_rng_get_state(const crec_ctx *ctx, uint32_t key[624], int32_t *pos)
{
    if (!ctx || !key || !pos)
        return CREC_EINVAL;
    std::memcpy(key, ctx->rng.key, sizeof(ctx->rng.key));
    *pos = ctx->rng.pos;
    return CREC_OK;
}

int crec_rng_randint(crec_ctx *ctx, int64_t n, int64_t *out)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, out && n >= 1 && n <= (int64_t)1 << 32, "randint range");
    uint32_t max = (uint32_t)(n - 1);
    *out = max == 0xffffffffu ? (int64_t)mt_next(ctx->rng) : (int64_t)draw(ctx->rng, max, mask_of(max));
    return CREC_OK;
}

int crec_rng_permutation(crec_ctx *ctx, int64_t n, int64_t *out)
{
    if (!ctx)
        return CREC_EINVAL;
    CREC_REQUIRE(ctx, n >= 0 && n <= 0xffffffffLL && (out || n == 0), "permutation size");
    permutation(ctx->rng, n, out);
    return CREC_OK;
}

int crec_sample_pairwise(crec_ctx *ctx, const int64_t *ptr, const int32_t *items, int64_t n_users,
                         const int32_t *user_ids, int32_t item_nums, int neg_ratio, int32_t *u, int32_t *i, int32_t *j,
                         int64_t *n_out)
{
    if (!ctx)
        return CREC_EINVAL;
    Csr c{ptr, items, n_users, user_ids, item_nums};
    if (int rc = check_csr(ctx, c, neg_ratio, u, i, j, n_out))
        return rc;
    const int64_t n_pos = ptr[n_users], n = n_pos * neg_ratio;
    CREC_REQUIRE(ctx, n <= 0xffffffffLL, "epoch too large for a 32-bit permutation");
    auto stamp = alloc<int32_t>(item_nums, true);
    auto neg = alloc<int32_t>(n);
    auto perm = alloc<uint32_t>(n);
    auto eu = edge_users(c);
    if (!stamp || !neg || !perm || !eu)
        return crec_fail(ctx, CREC_ENOMEM, "host allocation failed");
    static const bool timing = getenv("CREC_SAMPLER_TIMING") != nullptr; // phase times on stderr (tools/prof_sampler.py)
    struct rusage ru0, ru1;
    const int cpu0 = timing ? sched_getcpu() : -1;
    if (timing)
        getrusage(RUSAGE_THREAD, &ru0);
    const auto t0 = std::chrono::steady_clock::now();
    if (!draw_negatives(ctx->rng, c, neg_ratio, neg.get(), stamp.get()))
        return crec_fail(ctx, CREC_EINVAL, "a user has fewer than neg_ratio unseen items");
    const auto t1 = std::chrono::steady_clock::now();
    if (timing) { // a slow serial phase with no switches and CPU time = wall time is a slow hardware thread (a busy SMT sibling)
        getrusage(RUSAGE_THREAD, &ru1);
        fprintf(stderr, "  negatives phase: cpu %d -> %d, involuntary switches %ld, voluntary %ld, thread cpu time %.2f ms\n", cpu0, sched_getcpu(),
                ru1.ru_nivcsw - ru0.ru_nivcsw, ru1.ru_nvcsw - ru0.ru_nvcsw,
                (ru1.ru_utime.tv_sec - ru0.ru_utime.tv_sec) * 1e3 + (ru1.ru_utime.tv_usec - ru0.ru_utime.tv_usec) * 1e-3 +
                    (ru1.ru_stime.tv_sec - ru0.ru_stime.tv_sec) * 1e3 + (ru1.ru_stime.tv_usec - ru0.ru_stime.tv_usec) * 1e-3);
    }
    const int32_t *negp = neg.get(), *eup = eu.get();
    // sampler.py:68 + np.array(x)[s_idx] for the three columns at once
    const FastDiv by_ratio((uint32_t)neg_ratio);
    static const int ahead_mode = getenv("CREC_SAMPLER_AHEAD") ? atoi(getenv("CREC_SAMPLER_AHEAD")) : 0;
    shuffled_apply(
        ctx->rng, n, perm.get(),
        [=](int64_t k, uint32_t sidx) {
            const uint32_t e = by_ratio.div(sidx);
            u[k] = eup[e];
            i[k] = items[e];
            j[k] = negp[sidx];
        },
        [=](uint32_t sidx) {
            if (ahead_mode >= 1)
                __builtin_prefetch(&negp[sidx], 0, 0);
            if (ahead_mode >= 3) {
                const uint32_t e = by_ratio.div(sidx);
                __builtin_prefetch(&eup[e], 0, 0);
                __builtin_prefetch(&items[e], 0, 0);
            }
        });
    if (timing)
        fprintf(stderr, "crec_sample_pairwise: negatives %.2f ms, shuffle + gather %.2f ms\n",
                std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    *n_out = n;
    return CREC_OK;
}

int crec_sample_pointwise(crec_ctx *ctx, const int64_t *ptr, const int32_t *items, int64_t n_users,
                          const int32_t *user_ids, int32_t item_nums, int neg_ratio, int32_t *u, int32_t *i, float *y,
                          int64_t *n_out)
{
    if (!ctx)
        return CREC_EINVAL;
    Csr c{ptr, items, n_users, user_ids, item_nums};
    if (int rc = check_csr(ctx, c, neg_ratio, u, i, y, n_out))
        return rc;
    const int64_t n_pos = ptr[n_users], w = 1 + (int64_t)neg_ratio, n = n_pos * w;
    CREC_REQUIRE(ctx, n <= 0xffffffffLL, "epoch too large for a 32-bit permutation");
    auto stamp = alloc<int32_t>(item_nums, true);
    auto neg = alloc<int32_t>(n_pos * neg_ratio);
    auto perm = alloc<uint32_t>(n);
    auto eu = edge_users(c);
    if (!stamp || !neg || !perm || !eu)
        return crec_fail(ctx, CREC_ENOMEM, "host allocation failed");
    if (!draw_negatives(ctx->rng, c, neg_ratio, neg.get(), stamp.get()))
        return crec_fail(ctx, CREC_EINVAL, "a user has fewer than neg_ratio unseen items");
    const int32_t *negp = neg.get(), *eup = eu.get();
    const FastDiv by_w((uint32_t)w);
    static const int ahead_mode = getenv("CREC_SAMPLER_AHEAD") ? atoi(getenv("CREC_SAMPLER_AHEAD")) : 0;
    shuffled_apply(
        ctx->rng, n, perm.get(),
        [=](int64_t k, uint32_t sidx) { // sampler.py:37
            // instance s = e*(1+neg_ratio) + t: t == 0 the positive, t >= 1 its (t-1)-th negative (sampler.py:18-31)
            const uint32_t e = by_w.div(sidx), t = sidx - e * (uint32_t)w;
            u[k] = eup[e];
            i[k] = t == 0 ? items[e] : negp[(int64_t)e * neg_ratio + (t - 1)];
            y[k] = t == 0 ? 1.0f : 0.0f;
        },
        [=](uint32_t sidx) {
            if (ahead_mode < 1) // the same knob as the pairwise gather (CREC_SAMPLER_AHEAD): off unless asked for
                return;
            const uint32_t e = by_w.div(sidx), t = sidx - e * (uint32_t)w;
            __builtin_prefetch(&eup[e], 0, 0);
            __builtin_prefetch(t == 0 ? &items[e] : &negp[(int64_t)e * neg_ratio + (t - 1)], 0, 0);
        });
    *n_out = n;
    return CREC_OK;
}

// ranking_sampler_sbpr (utils/sampler.py:102-141).  Users without an SPu list are skipped (:106-107); per positive and
// repetition: s = randint(len(spu)) (no word drawn when the list has one entry), then neg = randint(item_nums) redrawn
// while it is a train item or an SPu item of the user (:113-119; no distinctness among the repetitions); the social
// coefficient of the drawn item is a property of (user, item) and comes precomputed (spu_suk); then one permutation.
int crec_sample_sbpr(crec_ctx *ctx, const int64_t *ptr, const int32_t *items, int64_t n_users, const int32_t *user_ids,
                     int32_t item_nums, int neg_ratio, const int64_t *spu_ptr, const int32_t *spu_items, const int32_t *spu_suk,
                     int32_t *u, int32_t *i, int32_t *i_s, int32_t *i_neg, float *suk, int64_t *n_out)
{
    if (!ctx)
        return CREC_EINVAL;
    Csr c{ptr, items, n_users, user_ids, item_nums};
    if (int rc = check_csr(ctx, c, neg_ratio, u, i, i_neg, n_out))
        return rc;
    CREC_REQUIRE(ctx, spu_ptr && i_s && suk && (spu_ptr[n_users] == 0 || (spu_items && spu_suk)), "null argument");
    int64_t n = 0;
    for (int64_t k = 0; k < n_users; ++k) {
        CREC_REQUIRE(ctx, spu_ptr[k + 1] >= spu_ptr[k], "spu_ptr not monotone at %lld", (long long)k);
        if (spu_ptr[k + 1] > spu_ptr[k])
            n += (ptr[k + 1] - ptr[k]) * neg_ratio;
    }
    for (int64_t e = 0; e < spu_ptr[n_users]; ++e)
        CREC_REQUIRE(ctx, spu_items[e] >= 0 && spu_items[e] < item_nums, "SPu item %d out of range", spu_items[e]);
    CREC_REQUIRE(ctx, n <= 0xffffffffLL, "epoch too large for a 32-bit permutation");
    auto stamp = alloc<int32_t>(item_nums, true);
    auto edge = alloc<int32_t>(n);   // instance -> position of its positive in `items`
    auto soc = alloc<int32_t>(n);    // instance -> position of its social item in spu_items
    auto neg = alloc<int32_t>(n);
    auto perm = alloc<uint32_t>(n);
    auto eu = edge_users(c);
    if (!stamp || !edge || !soc || !neg || !perm || !eu)
        return crec_fail(ctx, CREC_ENOMEM, "host allocation failed");
    const uint32_t max = (uint32_t)(item_nums - 1), mask = mask_of(max);
    int64_t out = 0;
    for (int64_t k = 0; k < n_users; ++k) {
        const int64_t ns = spu_ptr[k + 1] - spu_ptr[k];
        if (ns == 0)
            continue; // `if u not in SPu: continue`
        const int32_t tag = (int32_t)(k + 1);
        int64_t uniq = 0;
        for (int64_t e = ptr[k]; e < ptr[k + 1]; ++e) {
            uniq += (stamp[items[e]] != tag);
            stamp[items[e]] = tag; // tru
        }
        for (int64_t e = spu_ptr[k]; e < spu_ptr[k + 1]; ++e) {
            uniq += (stamp[spu_items[e]] != tag);
            stamp[spu_items[e]] = tag; // spu
        }
        if (ptr[k + 1] > ptr[k] && uniq >= item_nums)
            return crec_fail(ctx, CREC_EINVAL, "a user's train and social items cover the catalogue: no negative exists");
        const uint32_t smax = (uint32_t)(ns - 1), smask = mask_of(smax);
        for (int64_t e = ptr[k]; e < ptr[k + 1]; ++e) {
            for (int t = 0; t < neg_ratio; ++t) {
                edge[out] = (int32_t)e;
                soc[out] = (int32_t)(spu_ptr[k] + draw(ctx->rng, smax, smask));
                int32_t j;
                do {
                    j = (int32_t)draw(ctx->rng, max, mask);
                } while (stamp[j] == tag);
                neg[out] = j;
                ++out;
            }
        }
    }
    const int32_t *edgep = edge.get(), *socp = soc.get(), *negp = neg.get(), *eup = eu.get();
    shuffled_apply(ctx->rng, n, perm.get(), [=](int64_t k, uint32_t sidx) { // sampler.py:132-138
        u[k] = eup[edgep[sidx]];
        i[k] = items[edgep[sidx]];
        i_s[k] = spu_items[socp[sidx]];
        i_neg[k] = negp[sidx];
        suk[k] = (float)spu_suk[socp[sidx]];
    });
    *n_out = n;
    return CREC_OK;
}

int crec_sample_cml(crec_ctx *ctx, const int64_t *ptr, const int32_t *items, int64_t n_users, const int32_t *user_ids,
                    int32_t item_nums, int neg_ratio, int32_t *u, int32_t *i, int32_t *neg_out, int64_t *n_out)
{
    if (!ctx)
        return CREC_EINVAL;
    Csr c{ptr, items, n_users, user_ids, item_nums};
    if (int rc = check_csr(ctx, c, neg_ratio, u, i, neg_out, n_out))
        return rc;
    const int64_t n = ptr[n_users];
    CREC_REQUIRE(ctx, n <= 0xffffffffLL, "epoch too large for a 32-bit permutation");
    auto stamp = alloc<int32_t>(item_nums, true);
    auto neg = alloc<int32_t>(n * neg_ratio);
    auto perm = alloc<uint32_t>(n);
    auto eu = edge_users(c);
    if (!stamp || !neg || !perm || !eu)
        return crec_fail(ctx, CREC_ENOMEM, "host allocation failed");
    if (!draw_negatives(ctx->rng, c, neg_ratio, neg.get(), stamp.get()))
        return crec_fail(ctx, CREC_EINVAL, "a user has fewer than neg_ratio unseen items");
    const int32_t *negp = neg.get(), *eup = eu.get();
    shuffled_apply(ctx->rng, n, perm.get(), [=](int64_t k, uint32_t e) { // sampler.py:97
        u[k] = eup[e];
        i[k] = items[e];
        std::memcpy(neg_out + k * neg_ratio, negp + (int64_t)e * neg_ratio, sizeof(int32_t) * (size_t)neg_ratio);
    });
    *n_out = n;
    return CREC_OK;
}

} // extern "C"
