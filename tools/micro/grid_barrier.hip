// cost of a device-wide barrier inside a cooperative launch on MI355X: flat counter vs two-level tree, with and
// without the release/acquire fences a real multi-step kernel needs (tools for DESIGN.md; see bpr_train.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline void wait_ge(unsigned *p, unsigned v)
{
    while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < v) __builtin_amdgcn_s_sleep(1);
}
__global__ __launch_bounds__(256) void k(unsigned *bar, int iters, int mode, int fan, long long *out)
{
    const unsigned nb = gridDim.x, b = blockIdx.x;
    long long t0 = 0;
    if (b == 0 && threadIdx.x == 0) t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 1; it <= iters; ++it) {
        __syncthreads();
        if (threadIdx.x == 0) {
            if (mode & 4) __threadfence();
            if (mode & 16) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if ((mode & 3) == 0) { // flat
                __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                wait_ge(bar, (unsigned)it * nb);
            } else { // two-level: groups of `fan` workgroups, counters 64 B apart; root at bar[0]; release flag at bar[16]
                const unsigned g = b / fan, ng = (nb + fan - 1) / fan, gsz = (g + 1) * fan <= nb ? fan : nb - g * fan;
                unsigned *c1 = bar + 32 + 16 * g;
                const unsigned t = __hip_atomic_fetch_add(c1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t == (unsigned)it * gsz - 1) { // last of the group
                    const unsigned r = __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (r == (unsigned)it * ng - 1)
                        __hip_atomic_store(bar + 16, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                wait_ge(bar + 16, (unsigned)it);
            }
            if (mode & 4) __threadfence();
            if (mode & 8) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    if (b == 0 && threadIdx.x == 0) out[0] = __builtin_amdgcn_s_memrealtime() - t0;
}
int main()
{
    unsigned *bar; long long *out;
    (void)hipMalloc(&bar, 1 << 20); (void)hipMalloc(&out, 64);
    const int iters = 200;
    for (int grid : {256, 610}) for (int mode : {0, 1, 5, 9, 17}) for (int fan : {32}) {
        if ((mode & 3) == 0 && fan != 8) continue;
        (void)hipMemset(bar, 0, 1 << 20);
        void *args[] = {&bar, (void *)&iters, &mode, &fan, &out};
        hipError_t e = hipLaunchCooperativeKernel((const void *)k, dim3(grid), dim3(256), args, 0, 0);
        (void)hipDeviceSynchronize();
        long long h; (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
        printf("grid %d %s%s fan %2d: %.2f us per barrier (%s)\n", grid, (mode & 3) ? "tree" : "flat", (mode & 4) ? "+fences" : (mode & 8) ? "+acquire only" : (mode & 16) ? "+release only" : "", fan,
               h * 0.01 / iters, hipGetErrorString(e));
    }
    return 0;
}
