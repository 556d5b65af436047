// What s_memtime counts: a chain of N dependent v_fma_f32 (4 cycles each on a SIMD16 for a wave64, one wave alone on its SIMD)
// and a chain of N dependent v_mfma_f32_16x16x4_f32 (8 passes = 32 cycles each), bracketed by s_memtime / s_memrealtime.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/clock_probe.hip -o /tmp/clock_probe && /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ void probe(long long *out, float *sink, int n)
{
    float x = threadIdx.x * 1e-9f, a = 1.0000001f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < n; ++k)
        x = __builtin_fmaf(x, a, 1e-7f);
    asm volatile("" : "+v"(x));
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < n; ++k)
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, a, acc, 0, 0, 0);
    asm volatile("" : "+v"(acc));
    long long t2 = __builtin_amdgcn_s_memtime(), r2 = __builtin_amdgcn_s_memrealtime();
    f32x4 b0 = acc, b1 = acc;
    for (int k = 0; k < n; k += 2) {
        b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, a, b0, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x, b1, 0, 0, 0);
    }
    asm volatile("" : "+v"(b0), "+v"(b1));
    long long t3 = __builtin_amdgcn_s_memtime(), r3 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = t1 - t0, out[1] = r1 - r0, out[2] = t2 - t1, out[3] = r2 - r1, out[4] = t3 - t2, out[5] = r3 - r2;
    }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = x + acc[0] + b0[1] + b1[2];
}
int main()
{
    long long *out, h[6];
    float *sink;
    hipMalloc(&out, sizeof h);
    hipMalloc(&sink, 4 * 256 * 1024);
    const int n = 4096;
    for (int wg : {1, 256, 1024}) // 64-thread workgroups: one wave alone on a SIMD ... four per CU
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(probe, dim3(wg), dim3(64), 0, 0, out, sink, n);
            hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
            printf("wgs %4d: fma chain %.2f clk/op (%.2f ns/op), mfma chain %.2f clk/op (%.2f ns/op), two mfma chains %.2f clk/op (%.2f ns/op)\n", wg,
                   (double)h[0] / n, 10.0 * h[1] / n, (double)h[2] / n, 10.0 * h[3] / n, (double)h[4] / n, 10.0 * h[5] / n);
        }
    return 0;
}
