// micro-benchmark of the eval_sweep inner structure: chains of 32 dependent 32x32x2 f32 MFMAs with A from 32
// registers and B from ds_read_b128, accumulator handed over between chains
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;
__global__ __launch_bounds__(256) void k(long long *out, float *sink, int mode, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[128 * 68];
    for (int i = threadIdx.x; i < 128 * 68; i += blockDim.x) lds[i] = 0.001f * i;
    __syncthreads();
    const int lane = threadIdx.x & 63, row = lane & 31, h = lane >> 5;
    float a[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) a[q] = 0.5f + lane + q;
    f32x16 accA = {}, accB = {};
    float thr[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) thr[r] = 1e30f + r;
    unsigned hitsum = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            f32x16 &acc = blk ? accB : accA;
            const f32x16 &prv = blk ? accA : accB;
            const float *bp = lds + ((blk * 32 + row) * 68) + h * 32 + (it & 1) * 64 * 68;
            unsigned hit = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 b = *reinterpret_cast<const float4 *>(bp + 4 * q);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 0], b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 1], b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 2], b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 3], b.w, acc, 0, 0, 0);
                if (mode >= 1) {
#pragma unroll
                    for (int r = 2 * q; r < 2 * q + 2; ++r) hit |= (prv[r] > thr[r]) ? (1u << r) : 0u;
                }
            }
            if (mode >= 1 && __ballot(hit != 0)) hitsum += hit;
        }
        if (mode >= 2) __syncthreads();
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = accA[0] + accB[1] + hitsum;
}
int main()
{
    long long *d; float *s;
    (void)hipMalloc(&d, 8 * 1024); (void)hipMalloc(&s, 4 * 1024 * 1024);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) { k<<<256, 256>>>(d, s, mode, iters); (void)hipDeviceSynchronize(); }
        long long h; (void)hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("mode %d: %.1f clk per 32-MFMA block (ideal 2048)\n", mode, h / (2.0 * iters));
    }
    return 0;
}
