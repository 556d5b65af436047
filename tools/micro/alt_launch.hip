// what a launch costs when it differs from the one before it: a chain of dependent 15 MB table sweeps (the size of
// the ml-1m step's state), (i) one kernel, one grid; (ii) one kernel, the grid alternating between two sizes;
// (iii) two kernels (same code, two symbols) alternating, one grid; (iv) both alternating.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define SWEEP(name)                                                                                      \
    __global__ void name(const float4 *__restrict__ src, float4 *__restrict__ dst, int n4)              \
    {                                                                                                    \
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {      \
            float4 v = src[i];                                                                           \
            v.x += 1.f;                                                                                  \
            dst[i] = v;                                                                                  \
        }                                                                                                \
    }
SWEEP(k_a)
SWEEP(k_b)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int n4 = 15 * 1024 * 1024 / 16;
    float4 *a, *b; CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16));
    CK(hipMemset(a, 0, n4 * 16)); CK(hipMemset(b, 0, n4 * 16));
    const int N = 2048;
    for (int mode = 0; mode < 4; ++mode) {
        auto chain = [&]() {
            for (int i = 0; i < N; ++i) {
                const unsigned grid = (mode & 1) && (i & 1) ? 1187 : 1219;
                const float4 *s = (i & 1) ? b : a; float4 *d = (i & 1) ? a : b;
                if ((mode & 2) && (i & 1)) hipLaunchKernelGGL(k_b, dim3(grid), dim3(256), 0, st, s, d, n4);
                else hipLaunchKernelGGL(k_a, dim3(grid), dim3(256), 0, st, s, d, n4);
            }
        };
        chain(); CK(hipStreamSynchronize(st));
        const double t0 = now();
        for (int r = 0; r < 4; ++r) chain();
        CK(hipStreamSynchronize(st));
        printf("%-28s %-22s %.2f us per launch\n", (mode & 2) ? "two kernels alternating," : "one kernel,",
               (mode & 1) ? "grid 1219 / 1187:" : "grid 1219:", (now() - t0) / (4.0 * N) * 1e6);
    }
    return 0;
}
