// lone-wave issue rate vs active lanes (is a partially filled wave64 cheaper?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "ec.h"
using namespace reef;
__global__ void k_chain(fe256 *io, int iters) {
    fe a = fe_from_table(io[threadIdx.x]);
    fe b = a;
    for (int i = 0; i < iters; ++i) a = fe_mul<0>(a, b);
    io[threadIdx.x] = fe_to_table<0>(a);
}
int main() {
    fe256 *d; (void)hipMalloc(&d, 64 * sizeof(fe256)); (void)hipMemset(d, 1, 64 * sizeof(fe256));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int bd : {64, 64, 33, 32, 17, 16, 1}) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(bd), 0, 0, d, 20000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("lanes %2d: %.3f ms, %.1f ns per fe_mul\n", bd, ms, ms * 1e6 / 20000);
    }
    // multiple waves in the same block (4 SIMDs): per-wave latency should hold
    for (int bd : {128, 256, 512}) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d, 20000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        (void)bd;
    }
    return 0;
}
