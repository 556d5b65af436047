// Can the launch of step k+1 hide behind step k when the steps are fully dependent?  A chain of dependent 15 MB table sweeps (the ml-1m
// step's state: read 7.5 MB, write 7.5 MB, ping-pong), (0) one stream, plain kernels; (1) the same kernels with write-through stores /
// agent-scope loads on one stream (what the hand-off below needs, priced alone); (2) two streams alternating: kernel k+1 is dispatched
// while kernel k runs, its waves wait for k's workgroup counter (all of k's stores acknowledged at memory) before they touch the tables.
// The result is checked (every element = number of sweeps).  hipcc --offload-arch=gfx950 -O3 overlap_launch.hip -o overlap_launch
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rs(const void *p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0x7fffffff, 0x00020000); }

#ifndef SLEEP
#define SLEEP 4
#endif
template <int MODE>
__global__ __launch_bounds__(256) void sweep(const float4 *__restrict__ src, float4 *__restrict__ dst, int n4, int *done, int wait_for, const int *desc)
{
    const int d = desc[blockIdx.x & 1023]; // a descriptor load in front of the data, as the real step has
    if (MODE == 2) {
        if (threadIdx.x == 0) {
            int spins = 0;
            // the previous launch's last workgroup raises 16 flags (one cache line each): a sixteenth of the waiters polls each
            const int *flag = done + 32 * (1 + (blockIdx.x & 15));
            while ((int)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - wait_for) < 0 && ++spins < (1 << 20))
                __builtin_amdgcn_s_sleep(SLEEP);
        }
        __syncthreads();
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        float4 v;
        if (MODE == 0) v = src[i];
        else { const u4 x = __builtin_amdgcn_raw_buffer_load_b128(rs(src), i * 16, 0, 16); v = make_float4(__uint_as_float(x.x), __uint_as_float(x.y), __uint_as_float(x.z), __uint_as_float(x.w)); }
        v.x += 1.f + (float)d;
        if (MODE == 0) dst[i] = v;
        else { const u4 x = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}; __builtin_amdgcn_raw_buffer_store_b128(x, rs(dst), i * 16, 0, 16); }
    }
    if (MODE == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int mine = wait_for + (int)gridDim.x; // cumulative count of workgroups once this launch is complete
            if (__hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1 == mine)
                for (int f = 1; f <= 16; ++f)
                    __hip_atomic_store(done + 32 * f, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t st[2]; CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
    const int n4 = 15 * 1024 * 1024 / 2 / 16;
    float4 *a, *b; int *done, *desc;
    CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16)); CK(hipMalloc(&done, 4096)); CK(hipMalloc(&desc, 4096));
    CK(hipMemset(desc, 0, 4096));
    const int N = 2048, G = 1219;
    for (int mode = 0; mode < 3; ++mode) {
        CK(hipMemset(a, 0, n4 * 16)); CK(hipMemset(b, 0, n4 * 16)); CK(hipMemset(done, 0, 4096));
        CK(hipDeviceSynchronize());
        int launched = 0;
        auto chain = [&]() {
            for (int i = 0; i < N; ++i, ++launched) {
                const float4 *s = (launched & 1) ? b : a; float4 *d = (launched & 1) ? a : b;
                hipStream_t q = mode == 2 ? st[launched & 1] : st[0];
                if (mode == 0) hipLaunchKernelGGL(sweep<0>, dim3(G), dim3(256), 0, q, s, d, n4, done, 0, desc);
                if (mode == 1) hipLaunchKernelGGL(sweep<1>, dim3(G), dim3(256), 0, q, s, d, n4, done, 0, desc);
                if (mode == 2) hipLaunchKernelGGL(sweep<2>, dim3(G), dim3(256), 0, q, s, d, n4, done, launched * G, desc);
            }
        };
        chain(); CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int r = 0; r < 4; ++r) chain();
        CK(hipDeviceSynchronize());
        const double us = (now() - t0) / (4.0 * N) * 1e6;
        float4 h; CK(hipMemcpy(&h, (launched & 1) ? b : a, 16, hipMemcpyDeviceToHost));
        float4 h2; CK(hipMemcpy(&h2, ((launched & 1) ? b : a) + n4 - 1, 16, hipMemcpyDeviceToHost));
        printf("mode %d: %.2f us per launch; element 0 = %.0f, last = %.0f (want %d)\n", mode, us, h.x, h2.x, launched);
    }
    return 0;
}
