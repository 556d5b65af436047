// What a dataflow step of bpr_steps_flow can cost at best on MI355X (tools for DESIGN.md 4.2b):
//   A  dependent loads: plain (L2) vs agent-scope (sc1, served where the XCDs meet), small and large footprints
//   B  ping-pong between two waves on different / the same XCD through written-through stores and sc1 polls: one-way latency
//   C  the same with N pairs at once, each poll a 16-byte-per-lane row read (the polling storm) or a 4-byte flag
// hipcc --offload-arch=gfx950 -O3 -o flow_latency tools/micro/flow_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
typedef unsigned u4v __attribute__((ext_vector_type(4)));
__device__ inline unsigned ld_sc1(const unsigned *base, int byte_off)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(base), 0, 0x7fffffff, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 16);
}
__device__ inline u4v ld4_sc1(const unsigned *base, int byte_off)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(base), 0, 0x7fffffff, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16);
}
__device__ inline void st_wt(unsigned *base, int byte_off, unsigned v)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(v, r, byte_off, 0, 16);
}
__device__ inline void st4_wt(unsigned *base, int byte_off, u4v v)
{
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, byte_off, 0, 16);
}
__global__ void chase(const unsigned *next, int n, int sc1, long long *out)
{
    unsigned p = 0;
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int k = 0; k < n; ++k)
        p = sc1 ? ld_sc1(next, (int)p * 4) : next[p];
    out[0] = __builtin_amdgcn_s_memrealtime() - t0;
    out[1] = p;
}
// pairs of workgroups (a, b): a writes row version it, b polls it and answers, a polls the answer.  mode 0: 4-byte flags;
// mode 1: a 1 KB row per wave (16 bytes per lane), every word carries the iteration; mode 2: row + flag hint (poll the flag, then read the row)
__global__ __launch_bounds__(64) void pingpong(unsigned *buf, int iters, int mode, int partner_stride, int pairs, long long *out)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const int pair = b % pairs, side = (b / pairs) & 1; // blocks [0, pairs) are the a sides, [pairs, 2 pairs) the b sides: partner = b +- pairs
    (void)partner_stride;
    unsigned *mine = buf + (size_t)(2 * pair + side) * 4096, *theirs = buf + (size_t)(2 * pair + (1 - side)) * 4096; // 16 KB apart
    long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 1; it <= iters; ++it) {
        if (side == 0 || it > 0) {
            if (side == 1) { // wait for a's version `it`
                if (mode == 0) {
                    while (ld_sc1(theirs, 0) != (unsigned)it) __builtin_amdgcn_s_sleep(1);
                } else if (mode == 1) {
                    for (;;) { const u4v v = ld4_sc1(theirs, 64 + lane * 16); if (__ballot(v.x != (unsigned)it || v.w != (unsigned)it) == 0) break; __builtin_amdgcn_s_sleep(1); }
                } else {
                    while (ld_sc1(theirs, 0) != (unsigned)it) __builtin_amdgcn_s_sleep(1);
                    for (;;) { const u4v v = ld4_sc1(theirs, 64 + lane * 16); if (__ballot(v.x != (unsigned)it || v.w != (unsigned)it) == 0) break; }
                }
            }
            // publish my version `it`
            if (mode >= 1) { const u4v v = {(unsigned)it, (unsigned)it, (unsigned)it, (unsigned)it}; st4_wt(mine, 64 + lane * 16, v); }
            if (mode != 1 && lane == 0) st_wt(mine, 0, (unsigned)it);
            if (side == 0) { // wait for b's answer
                if (mode == 0) {
                    while (ld_sc1(theirs, 0) != (unsigned)it) __builtin_amdgcn_s_sleep(1);
                } else if (mode == 1) {
                    for (;;) { const u4v v = ld4_sc1(theirs, 64 + lane * 16); if (__ballot(v.x != (unsigned)it || v.w != (unsigned)it) == 0) break; __builtin_amdgcn_s_sleep(1); }
                } else {
                    while (ld_sc1(theirs, 0) != (unsigned)it) __builtin_amdgcn_s_sleep(1);
                    for (;;) { const u4v v = ld4_sc1(theirs, 64 + lane * 16); if (__ballot(v.x != (unsigned)it || v.w != (unsigned)it) == 0) break; }
                }
            }
        }
    }
    if (lane == 0) out[b] = __builtin_amdgcn_s_memrealtime() - t0;
}
int main()
{
    long long *out; (void)hipMalloc(&out, 8 * 8192);
    for (size_t n : {(size_t)1 << 12, (size_t)1 << 21, (size_t)1 << 26}) { // 16 KB, 8 MB, 256 MB
        std::vector<unsigned> perm(n), next(n);
        std::iota(perm.begin(), perm.end(), 0u);
        std::mt19937 g(1);
        std::shuffle(perm.begin() + 1, perm.end(), g);
        for (size_t k = 0; k + 1 < n; ++k) next[perm[k]] = perm[k + 1];
        next[perm[n - 1]] = perm[0];
        unsigned *d; (void)hipMalloc(&d, n * 4);
        (void)hipMemcpy(d, next.data(), n * 4, hipMemcpyHostToDevice);
        for (int sc1 : {0, 1}) {
            const int steps = 20000;
            hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, d, steps, sc1, out);
            hipLaunchKernelGGL(chase, dim3(1), dim3(1), 0, 0, d, steps, sc1, out);
            (void)hipDeviceSynchronize();
            long long h[2]; (void)hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
            printf("chase %8zu KB %s: %.0f ns per load\n", n * 4 / 1024, sc1 ? "sc1  " : "plain", h[0] * 10.0 / steps);
        }
        (void)hipFree(d);
    }
    unsigned *buf; (void)hipMalloc(&buf, (size_t)4096 * 4 * 2 * 2048);
    for (int mode : {0, 1, 2}) for (int pairs : {1, 8, 9, 256, 257, 1025}) {
        // pairs == 8: a and b of a pair on the SAME XCD (blocks b and b + 8); 1, 9: different XCDs
        (void)hipMemset(buf, 0, (size_t)4096 * 4 * 2 * 2048);
        const int iters = 2000;
        hipLaunchKernelGGL(pingpong, dim3(2 * pairs), dim3(64), 0, 0, buf, iters, mode, 0, pairs, out);
        (void)hipDeviceSynchronize();
        std::vector<long long> h(2 * pairs); (void)hipMemcpy(h.data(), out, 8 * 2 * pairs, hipMemcpyDeviceToHost);
        const long long worst = *std::max_element(h.begin(), h.end());
        printf("pingpong %s, %4d pairs (%s): %.0f ns one way (slowest pair)\n", mode == 0 ? "4 B flags      " : mode == 1 ? "1 KB rows      " : "rows + flag hint", pairs,
               pairs % 8 == 0 ? "same XCD" : "other XCD", worst * 10.0 / iters / 2);
    }
    return 0;
}
