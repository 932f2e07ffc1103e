// hoststep_probe -- what does a step on a HOST core cost when it sits between two pieces of stream work?  (round 6: the window combine of a plain
// key whose result stays on the device is such a step -- engine.inc, HostCombineJob -- and ships as form A.)
//   A  hipLaunchHostFunc(fn) + hipMemcpyAsync(96 B): what the library does
//   B  a helper thread that sleeps in hipEventSynchronize on an event recorded BEFORE the last kernel, then polls a flag the kernel writes to host
//      memory, does the step, and releases the stream with a flag hipStreamWaitValue32 waits for; then hipMemcpyAsync(96 B)
//   C  as B, but the helper polls from the moment the work is posted (a core burnt for the whole MSM)
// The "last kernel" spins for tail_us (the bucket reduction of an MSM), the step itself burns step_us on the host (the 255 doublings: ~100 us).
// Printed per form: median over reps of (launch -> result on the device, host-synchronised) minus the kernel's own duration minus the step.
// usage: hoststep_probe [tail_us=300] [step_us=100] [reps=200]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
using clk = std::chrono::steady_clock;
static double us_since(clk::time_point a) { return std::chrono::duration<double, std::micro>(clk::now() - a).count(); }
static void burn(double us) { const auto t = clk::now(); while (us_since(t) < us) {} }
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// spins for `ticks` of the 100 MHz wall clock, then (flag != nullptr) publishes seq to host memory
__global__ void k_tail(unsigned long long ticks, volatile unsigned *flag, unsigned seq, unsigned *sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0) {
        sink[0] = seq;
        if (flag) { __threadfence_system(); *flag = seq; }
    }
}

struct Job { double step_us; unsigned char *result; };
static void host_fn(void *p) {
    Job *j = (Job *)p;
    burn(j->step_us);
    memset(j->result, 0x5a, 96);
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char **argv) {
    const double tail_us = argc > 1 ? atof(argv[1]) : 300, step_us = argc > 2 ? atof(argv[2]) : 100;
    const int reps = argc > 3 ? atoi(argv[3]) : 200;
    hipStream_t s;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned *sink; unsigned char *dst, *pinned; unsigned *flags;       // flags[0]: kernel -> host ("ready"), flags[16]: host -> stream ("done")
    CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&dst, 96));
    CHECK(hipHostMalloc(&pinned, 128, hipHostMallocDefault));
    CHECK(hipHostMalloc(&flags, 256, hipHostMallocDefault));
    memset(flags, 0, 256);
    unsigned *flags_dev = nullptr;
    CHECK(hipHostGetDevicePointer((void **)&flags_dev, flags, 0));
    int can_wait = 0;
    (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0);
    const unsigned long long ticks = (unsigned long long)(tail_us * 100.0);
    hipEvent_t e0, e1, early;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreateWithFlags(&early, hipEventBlockingSync | hipEventDisableTiming));
    // the kernel's own duration (events) and the floor: kernel + 96-B copy, no host step
    std::vector<double> kd, floor_us;
    for (int i = 0; i < 20 + reps; ++i) {
        const auto t = clk::now();
        CHECK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(k_tail, dim3(1), dim3(64), 0, s, ticks, (volatile unsigned *)nullptr, 0u, sink);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipMemcpyAsync(dst, pinned, 96, hipMemcpyHostToDevice, s));
        CHECK(hipStreamSynchronize(s));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 20) { kd.push_back(ms * 1e3); floor_us.push_back(us_since(t)); }
    }
    const double kernel_us = median(kd);
    printf("tail kernel %.1f us (asked %.0f); kernel + 96-B copy + sync, no host step: %.1f us; the step itself %.0f us; hipStreamWaitValue32 usable: %d\n", kernel_us, tail_us,
           median(floor_us), step_us, can_wait);
    Job job{step_us, pinned};
    // A
    {
        std::vector<double> v;
        for (int i = 0; i < 20 + reps; ++i) {
            const auto t = clk::now();
            hipLaunchKernelGGL(k_tail, dim3(1), dim3(64), 0, s, ticks, (volatile unsigned *)nullptr, 0u, sink);
            CHECK(hipLaunchHostFunc(s, host_fn, &job));
            CHECK(hipMemcpyAsync(dst, pinned, 96, hipMemcpyHostToDevice, s));
            CHECK(hipStreamSynchronize(s));
            if (i >= 20) v.push_back(us_since(t));
        }
        printf("A  hipLaunchHostFunc:                      %.1f us in all = %.1f us beyond kernel + step\n", median(v), median(v) - kernel_us - step_us);
    }
    if (!can_wait) { printf("B, C need hipStreamWaitValue32\n"); return 0; }
    // B and C: the helper thread
    for (int form = 0; form < 2; ++form) {
        std::mutex mu; std::condition_variable cv;
        unsigned posted = 0; bool quit = false;
        std::atomic<double> spun_us{0};
        volatile unsigned *ready = flags, *done = flags + 16;
        std::thread helper([&] {
            unsigned seen = 0;
            for (;;) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return quit || posted != seen; });
                    if (quit) return;
                    seen = posted;
                }
                if (form == 0) (void)hipEventSynchronize(early);            // asleep until the work BEFORE the tail kernel has finished
                const auto t = clk::now();
                while (*ready != seen) {}                                       // the tail kernel's flag: polled for about the tail's length
                spun_us.store(spun_us.load() + us_since(t));
                burn(step_us);
                memset(pinned, 0x5a, 96);
                std::atomic_thread_fence(std::memory_order_release);
                *done = seen;                                                   // releases the stream
            }
        });
        std::vector<double> v;
        for (int i = 0; i < 20 + reps; ++i) {
            const unsigned seq = (unsigned)(i + 1) + (form ? 100000u : 0u);
            const auto t = clk::now();
            CHECK(hipEventRecord(early, s));
            {
                std::lock_guard<std::mutex> lk(mu);
                posted = seq;
            }
            cv.notify_one();
            hipLaunchKernelGGL(k_tail, dim3(1), dim3(64), 0, s, ticks, (volatile unsigned *)flags_dev, seq, sink);
            CHECK(hipStreamWaitValue32(s, (void *)(flags_dev + 16), seq, hipStreamWaitValueEq, 0xffffffffu));
            CHECK(hipMemcpyAsync(dst, pinned, 96, hipMemcpyHostToDevice, s));
            CHECK(hipStreamSynchronize(s));
            if (i >= 20) v.push_back(us_since(t));
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv.notify_one();
        helper.join();
        printf("%s %.1f us in all = %.1f us beyond kernel + step; the helper polled %.0f us per call\n",
               form == 0 ? "B  helper asleep on an early event, then polls:" : "C  helper polls from the post:               ", median(v), median(v) - kernel_us - step_us,
               spun_us.load() / (20 + reps));
    }
    return 0;
}
