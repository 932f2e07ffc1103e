// How long does the host wait for a tiny kernel?  hipStreamSynchronize against polling a host-mapped word the kernel
// writes last (system-scope release).  Decides whether the latency-bound rounds (sum-check, IPA) should wait by polling.
//   hipcc --offload-arch=gfx950 -O2 sync_probe.hip -o sync_probe && ./sync_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>

__global__ void k_tiny(volatile uint32_t *flag, uint32_t seq, uint32_t *sink) {
    if (threadIdx.x == 0) {
        sink[0] = seq;
        __threadfence_system();
        *flag = seq;
    }
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { printf("no GPU\n"); return 1; }
    uint32_t *flag = nullptr, *sink = nullptr;
    hipHostMalloc((void **)&flag, 64, hipHostMallocDefault);
    hipMalloc((void **)&sink, 64);
    *flag = 0;
    const int iters = 2000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now_us();
            for (int i = 1; i <= iters; ++i) {
                const uint32_t seq = (uint32_t)(mode * 100000 + rep * 10000 + i);
                hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, flag, seq, sink);
                if (mode == 0) {
                    hipStreamSynchronize(st);
                } else {
                    while (*(volatile uint32_t *)flag != seq) { }
                }
            }
            double dt = (now_us() - t0) / iters;
            hipStreamSynchronize(st);
            if (rep == 1) printf("%s: %.2f us per launch + wait\n", mode == 0 ? "hipStreamSynchronize" : "poll host-mapped flag", dt);
        }
    }
    // two dependent tiny kernels then a wait (the shape of a round before the one-launch form)
    for (int mode = 0; mode < 2; ++mode) {
        double t0 = now_us();
        for (int i = 1; i <= iters; ++i) {
            const uint32_t seq = (uint32_t)(500000 + mode * 10000 + i);
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, flag + 8, seq, sink);
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, flag, seq, sink);
            if (mode == 0) hipStreamSynchronize(st);
            else while (*(volatile uint32_t *)flag != seq) { }
        }
        printf("two kernels, %s: %.2f us\n", mode == 0 ? "hipStreamSynchronize" : "poll", (now_us() - t0) / iters);
        hipStreamSynchronize(st);
    }
    return 0;
}
