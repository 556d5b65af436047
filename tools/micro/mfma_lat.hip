// micro-benchmark: dependent / independent chains of f32 MFMA, with and without LDS operand reads
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ void k(long long *out, float *sink, int mode)
{
    __shared__ float lds[64 * 66];
    for (int i = threadIdx.x; i < 64 * 66; i += blockDim.x) lds[i] = 0.001f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, l16 = lane & 15, q = lane >> 4;
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    f32x16 c0 = {};
    float x = 0.5f + lane, y = 0.25f;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 0) {
#pragma unroll 1
        for (int it = 0; it < 64; ++it) {
#pragma unroll
            for (int s = 0; s < 16; ++s) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        }
    } else if (mode == 1) {
#pragma unroll 1
        for (int it = 0; it < 64; ++it) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
            }
        }
    } else if (mode == 2) {
#pragma unroll 1
        for (int it = 0; it < 64; ++it) {
#pragma unroll
            for (int s = 0; s < 16; ++s) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c0, 0, 0, 0);
        }
    } else if (mode == 3) { // 16 reads (a: k along row stride 65; b: k down the column) then 16 MFMA
#pragma unroll 1
        for (int it = 0; it < 64; ++it) {
            float a[16], b[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                a[s] = lds[l16 * 65 + ((it & 1) * 0 + s + 16 * q)];
                b[s] = lds[(s + 16 * q) * 65 + l16 + (it & 1)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 16; ++s) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], a0, 0, 0, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + c0[3];
}
int main()
{
    long long *d; float *s;
    hipMalloc(&d, 8 * 1024); hipMalloc(&s, 4 * 1024 * 1024);
    for (int nt : {64, 256, 512}) for (int mode = 0; mode < 4; ++mode) {
        k<<<1, nt>>>(d, s, mode); hipDeviceSynchronize();
        k<<<1, nt>>>(d, s, mode); hipDeviceSynchronize();
        long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("threads %d mode %d: %lld clk for 1024 mfma per wave -> %.1f clk/mfma\n", nt, mode, h, h / 1024.0);
    }
    return 0;
}
