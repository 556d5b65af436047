// Cost of __syncthreads() in a 512-thread workgroup (8 waves), of a dependent LDS read, and of a small "phase" the way
// csrc/ncf.hip's tile has them (two waves work, six go straight to the barrier), in s_memtime clocks.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/barrier_probe.hip -o /tmp/barrier_probe && /tmp/barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ __launch_bounds__(512) void probe(long long *out, float *sink, int lds_bytes)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int k = tid; k < lds_bytes / 4; k += 512)
        lds[k] = (float)((k * 7 + 1) % 1024);
    __syncthreads();
    long long t[6];
    t[0] = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int k = 0; k < 64; ++k)
        __syncthreads();
    t[1] = __builtin_amdgcn_s_memtime();
    int idx = tid;
#pragma unroll 1
    for (int k = 0; k < 64; ++k) // dependent LDS reads
        idx = (int)lds[idx & 1023];
    t[2] = __builtin_amdgcn_s_memtime();
    float x = (float)idx;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#ifdef STRAIGHT // the same 64 phases as straight-line code, each executed once (the way a tile runs its phases)
#pragma unroll
#else
#pragma unroll 1
#endif
    for (int k = 0; k < 64; ++k) { // a small phase: waves 0, 1 read 16 operands, 8 MFMAs, write 4 results; all meet at a barrier
        if (wave < 2) {
            float a[8], b[8];
#pragma unroll
            for (int s = 0; s < 8; ++s)
                a[s] = lds[(tid & 15) * 33 + s + 8 * ((tid & 63) >> 4)], b[s] = lds[2048 + (s + 8 * ((tid & 63) >> 4)) * 17 + (tid & 15)];
#pragma unroll
            for (int s = 0; s < 8; ++s)
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                lds[4096 + (4 * ((tid & 63) >> 4) + j) * 17 + (tid & 15) + wave * 512] = acc[j] + x;
        }
        __syncthreads();
    }
    t[3] = __builtin_amdgcn_s_memtime();
    if (tid == 0 && blockIdx.x == 0)
        for (int k = 0; k < 3; ++k)
            out[k] = t[k + 1] - t[k];
    sink[blockIdx.x * 512 + tid] = x + acc[0] + lds[tid];
}
int main()
{
    long long *out, h[3];
    float *sink;
    hipMalloc(&out, sizeof h);
    hipMalloc(&sink, 4 * 512 * 1024);
    for (int lds_kb : {32, 108})
        for (int wg : {1, 192}) {
            hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
            for (int rep = 0; rep < 2; ++rep) {
                hipLaunchKernelGGL(probe, dim3(wg), dim3(512), lds_kb * 1024, 0, out, sink, lds_kb * 1024);
                if (hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost) != hipSuccess)
                    return 1;
                printf("lds %3d KB, %3d workgroups: barrier %.0f clk, dependent LDS read %.0f clk, small phase %.0f clk\n", lds_kb, wg, h[0] / 64.0,
                       h[1] / 64.0, h[2] / 64.0);
            }
        }
    return 0;
}
