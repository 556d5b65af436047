// dispatch + drain floor of an empty kernel as a function of grid and block size, measured like the library does:
// start/stop events attached to the dispatch (hipExtLaunchKernelGGL), back-to-back launches on one stream
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
struct Args { void *p[24]; };
__global__ void k_empty(Args a, int n) { if (n < 0) ((int *)a.p[0])[0] = 1; }
int main()
{
    hipStream_t st; (void)hipStreamCreate(&st);
    const int N = 200;
    hipEvent_t a[N], b[N];
    for (int i = 0; i < N; ++i) { (void)hipEventCreate(&a[i]); (void)hipEventCreate(&b[i]); }
    Args args{};
    for (int block : {256, 1024}) for (int grid : {1, 64, 256, 610, 1219, 2438, 4876}) {
        for (int i = 0; i < N; ++i) hipExtLaunchKernelGGL(k_empty, dim3(grid), dim3(block), 0, st, a[i], b[i], 0, args, 0);
        (void)hipStreamSynchronize(st);
        double tot = 0; float ms;
        for (int i = 20; i < N; ++i) { (void)hipEventElapsedTime(&ms, a[i], b[i]); tot += ms; }
        float span; (void)hipEventElapsedTime(&span, a[20], b[N - 1]);
        printf("block %4d grid %5d: %.2f us per dispatch (begin-end), %.2f us per launch back-to-back\n", block, grid, tot / (N - 20) * 1e3, span / (N - 20) * 1e3);
    }
    return 0;
}
