// What does the memory system give a ONE-LAUNCH-PER-BATCH step with the S-big access pattern, arithmetic aside?
// Per triple: 9 random 512-byte row slices read (3 rows x {var, m, v}), 9 written back; 6144 triples per launch; tables far beyond the
// Infinity Cache.  Variants: (0) ids from memory, one round per wave (two triples per wave, 32 lanes x 16 B per row) -- the shape of
// bpr_triples_lazy's main line; (1) ids computed in the kernel (no dependent descriptor load); (2) reads only; (3) writes only;
// (4) four triples per wave in two rounds, the second round's loads issued before the first round's stores; (5) like 0 with the
// three slots of a row interleaved in one 1536-byte record.   hipcc --offload-arch=gfx950 -O3 gather_update_floor.hip -o floor && ./floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int D = 128, B = 6144, NB = 64;
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__device__ __forceinline__ f4 upd(f4 a, f4 b, f4 c) { return a * 0.999f + b * 0.001f + c; }

template <int V>
__global__ __launch_bounds__(512) void step(float *P, float *mP, float *vP, float *Q, float *mQ, float *vQ, const int *ids, int batch, uint32_t U, uint32_t I)
{
    const int lane = threadIdx.x & 31, g = (blockIdx.x * 512 + threadIdx.x) >> 5; // group = triple
    constexpr int ROUNDS = V == 4 ? 2 : 1;
    const int per = B / ROUNDS;
    int64_t o[ROUNDS][3];
    f4 x[ROUNDS][9];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int t = g + r * per;
        uint32_t u, i, j;
        if (V == 1) { u = mix(t * 3 + batch * 77777) % U; i = mix(t * 3 + 1 + batch * 77777) % I; j = mix(t * 3 + 2 + batch * 77777) % I; }
        else { const int *p = ids + ((int64_t)batch * B + t) * 3; u = p[0]; i = p[1]; j = p[2]; }
        const int S = V == 5 ? 3 * D : D;
        o[r][0] = (int64_t)u * S + lane * 4; o[r][1] = (int64_t)i * S + lane * 4; o[r][2] = (int64_t)j * S + lane * 4;
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        if (V == 3) { for (int k = 0; k < 9; ++k) x[r][k] = (f4){1.f, 2.f, 3.f, (float)lane}; continue; }
        if (V == 5) {
            for (int k = 0; k < 3; ++k) { x[r][k] = *(f4 *)(P + o[r][0] + k * D); x[r][3 + k] = *(f4 *)(Q + o[r][1] + k * D); x[r][6 + k] = *(f4 *)(Q + o[r][2] + k * D); }
        } else {
            x[r][0] = *(f4 *)(P + o[r][0]); x[r][1] = *(f4 *)(mP + o[r][0]); x[r][2] = *(f4 *)(vP + o[r][0]);
            x[r][3] = *(f4 *)(Q + o[r][1]); x[r][4] = *(f4 *)(mQ + o[r][1]); x[r][5] = *(f4 *)(vQ + o[r][1]);
            x[r][6] = *(f4 *)(Q + o[r][2]); x[r][7] = *(f4 *)(mQ + o[r][2]); x[r][8] = *(f4 *)(vQ + o[r][2]);
        }
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        f4 y[9];
        for (int k = 0; k < 9; ++k) y[k] = upd(x[r][k], x[r][(k + 3) % 9], x[r][(k + 6) % 9]);
        if (V == 2) { f4 s = y[0]; for (int k = 1; k < 9; ++k) s += y[k]; if (s.x == 12345.678f) P[o[r][0]] = s.y; continue; }
        if (V == 5) {
            for (int k = 0; k < 3; ++k) { __builtin_nontemporal_store(y[k], (f4 *)(P + o[r][0] + k * D)); __builtin_nontemporal_store(y[3 + k], (f4 *)(Q + o[r][1] + k * D)); __builtin_nontemporal_store(y[6 + k], (f4 *)(Q + o[r][2] + k * D)); }
        } else {
            __builtin_nontemporal_store(y[0], (f4 *)(P + o[r][0])); __builtin_nontemporal_store(y[1], (f4 *)(mP + o[r][0])); __builtin_nontemporal_store(y[2], (f4 *)(vP + o[r][0]));
            __builtin_nontemporal_store(y[3], (f4 *)(Q + o[r][1])); __builtin_nontemporal_store(y[4], (f4 *)(mQ + o[r][1])); __builtin_nontemporal_store(y[5], (f4 *)(vQ + o[r][1]));
            __builtin_nontemporal_store(y[6], (f4 *)(Q + o[r][2])); __builtin_nontemporal_store(y[7], (f4 *)(mQ + o[r][2])); __builtin_nontemporal_store(y[8], (f4 *)(vQ + o[r][2]));
        }
    }
}

template <int V> float run(float **t, const int *ids, uint32_t U, uint32_t I)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int groups = V == 4 ? B / 2 : B, blocks = groups * 32 / 512;
    for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(step<V>, dim3(blocks), dim3(512), 0, 0, t[0], t[1], t[2], t[3], t[4], t[5], ids, k, U, I);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    for (int k = 0; k < NB; ++k) hipLaunchKernelGGL(step<V>, dim3(blocks), dim3(512), 0, 0, t[0], t[1], t[2], t[3], t[4], t[5], ids, k, U, I);
    CK(hipEventRecord(b, 0));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3f / NB;
}

int main(int argc, char **argv)
{
    const uint32_t U = argc > 1 ? atoi(argv[1]) : 10000000, I = argc > 2 ? atoi(argv[2]) : 1000000;
    float *t[6];
    for (int k = 0; k < 3; ++k) { CK(hipMalloc(&t[k], (size_t)U * D * 4)); CK(hipMemset(t[k], 0, (size_t)U * D * 4)); }
    for (int k = 3; k < 6; ++k) { CK(hipMalloc(&t[k], (size_t)I * D * 4 * (k == 3 ? 3 : 1))); CK(hipMemset(t[k], 0, (size_t)I * D * 4)); }
    std::vector<int> h((size_t)NB * B * 3);
    uint64_t s = 88172645463325252ull;
    for (size_t k = 0; k < h.size(); ++k) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[k] = (int)(s % (k % 3 == 0 ? U / 3 : I)); } // (U / 3: variant 5 strides P by 3 rows)
    int *ids; CK(hipMalloc(&ids, h.size() * 4)); CK(hipMemcpy(ids, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const double mb = B * 18.0 * 512 / 1e6;
    printf("U=%u I=%u d=%d B=%d: %.1f MB read+written per launch (half for variants 2, 3)\n", U, I, D, B, mb);
    float us;
    us = run<0>(t, ids, U / 3, I); printf("0 ids from memory, one round        %6.2f us  %5.2f TB/s\n", us, mb / us);
    us = run<1>(t, ids, U / 3, I); printf("1 ids computed                      %6.2f us  %5.2f TB/s\n", us, mb / us);
    us = run<2>(t, ids, U / 3, I); printf("2 reads only                        %6.2f us  %5.2f TB/s\n", us, mb / 2 / us);
    us = run<3>(t, ids, U / 3, I); printf("3 writes only                       %6.2f us  %5.2f TB/s\n", us, mb / 2 / us);
    us = run<4>(t, ids, U / 3, I); printf("4 two rounds per wave, overlapped   %6.2f us  %5.2f TB/s\n", us, mb / us);
    us = run<5>(t, ids, U / 3, I / 3); printf("5 slots interleaved (1536-B rows)   %6.2f us  %5.2f TB/s\n", us, mb / us);
    return 0;
}
