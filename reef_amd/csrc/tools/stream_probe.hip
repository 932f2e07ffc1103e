// stream_probe -- what does hipStreamCreate cost, and whom does it stall?  Thread B creates streams one by one (timed); thread A meanwhile runs a
// tiny kernel + hipStreamSynchronize in a loop on its own stream and records the longest iteration seen during each creation.  Also: hipHostMalloc,
// hipMalloc and hipEventCreate from B.  (round 6: the drop-in symbols' builder thread creates streams and contexts beside the caller.)
// usage: GPU_MAX_HW_QUEUES=8 stream_probe
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
using clk = std::chrono::steady_clock;
static double ms(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }
__global__ void k_tiny(int *p) { if (threadIdx.x == 0) p[0] += 1; }
int main() {
    int *d = nullptr;
    hipMalloc(&d, 64);
    hipStream_t sa;
    auto t0 = clk::now();
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    printf("first stream (main thread): %.3f ms\n", ms(t0));
    for (int i = 0; i < 20; ++i) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, sa, d); hipStreamSynchronize(sa); }
    std::atomic<bool> stop{false};
    std::atomic<double> worst{0};
    std::thread a([&] {
        while (!stop.load()) {
            auto t = clk::now();
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, sa, d);
            hipStreamSynchronize(sa);
            const double v = ms(t);
            double w = worst.load();
            while (v > w && !worst.compare_exchange_weak(w, v)) {}
        }
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    printf("caller loop alone: worst iteration %.3f ms\n", worst.exchange(0));
    hipStream_t s[10];
    for (int i = 0; i < 10; ++i) {
        worst.store(0);
        t0 = clk::now();
        hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
        const double c = ms(t0);
        // first use of the stream: where the hardware queue is really created?
        t0 = clk::now();
        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s[i], d + 8);
        hipStreamSynchronize(s[i]);
        const double u = ms(t0);
        printf("stream %d: create %.3f ms, first launch+sync %.3f ms; caller's worst iteration meanwhile %.3f ms\n", i + 2, c, u, worst.load());
    }
    for (const char *what : {"hipHostMalloc 128 B", "hipMalloc 64 MiB", "hipEventCreate", "hipFree 64 MiB"}) {
        static void *p64 = nullptr;
        worst.store(0);
        t0 = clk::now();
        void *p = nullptr;
        hipEvent_t e;
        if (what[3] == 'H') hipHostMalloc(&p, 128, hipHostMallocDefault);
        else if (what[3] == 'M') hipMalloc(&p64, 64 << 20);
        else if (what[3] == 'E') hipEventCreate(&e);
        else hipFree(p64);
        printf("%s: %.3f ms; caller's worst iteration meanwhile %.3f ms\n", what, ms(t0), worst.load());
    }
    stop.store(true);
    a.join();
    return 0;
}
