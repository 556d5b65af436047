// per-kernel cost of a chain of dependent launches: plain stream launches against the same chain captured into a
// hipGraph and replayed (does a graph shorten the kernel boundary?).  Kernels: empty, and a 15 MB table sweep
// (the size of the ml-1m step's state) in 4876 waves like bpr_step_fused.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_empty(float *p, int n) { if (n < 0) p[0] = 1; }
__global__ void k_sweep(const float4 *__restrict__ src, float4 *__restrict__ dst, int n4)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        float4 v = src[i]; v.x += 1.f; dst[i] = v;
    }
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int n4 = 15 * 1024 * 1024 / 16;
    float4 *a, *b; CK(hipMalloc(&a, n4 * 16)); CK(hipMalloc(&b, n4 * 16));
    CK(hipMemset(a, 0, n4 * 16)); CK(hipMemset(b, 0, n4 * 16));
    for (int kind = 0; kind < 2; ++kind) for (int N : {64, 256, 1024}) {
        auto chain = [&](hipStream_t s) {
            for (int i = 0; i < N; ++i) {
                if (kind == 0) hipLaunchKernelGGL(k_empty, dim3(1219), dim3(256), 0, s, (float *)a, 0);
                else hipLaunchKernelGGL(k_sweep, dim3(1219), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, n4);
            }
        };
        chain(st); CK(hipStreamSynchronize(st));
        double t0 = now();
        for (int r = 0; r < 8; ++r) chain(st);
        CK(hipStreamSynchronize(st));
        double plain = (now() - t0) / (8.0 * N) * 1e6;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        chain(st);
        CK(hipStreamEndCapture(st, &g));
        double ti = now();
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double inst = (now() - ti) * 1e6;
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        t0 = now();
        for (int r = 0; r < 8; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        double graph = (now() - t0) / (8.0 * N) * 1e6;
        printf("%s chain of %4d: stream %.2f us/kernel, graph %.2f us/kernel (instantiate %.0f us = %.2f us/node)\n",
               kind ? "15MB-sweep" : "empty     ", N, plain, graph, inst, inst / N);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
