// Round trip host -> resident kernel -> host through host-mapped memory on MI355X: how cheap can a "mailbox" round be?
// The kernel polls a sequence word in pinned host memory and answers in another; variants of the poll and of the answer.
// hipcc --offload-arch=gfx950 -O2 mailbox_probe.hip -o mailbox_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned int u32;
typedef u32 v4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4 load_sys(const void *p) {
    v4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void store_sys(void *p, u32 v) {
    asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : : "v"(p), "v"(v) : "memory");
}
// mode bit 0: answer with __threadfence_system() + volatile store (0) or an sc0 sc1 store alone (1); bit 1: poll with volatile dword (0) or b128 asm (1)
__global__ void k_pingpong(const u32 *mail, u32 *ack, u32 rounds, int mode, u32 *dummy) {
    if (threadIdx.x != 0) return;
    for (u32 i = 1; i <= rounds; ++i) {
        unsigned long long t0 = wall_clock64();
        for (;;) {
            u32 s = (mode & 2) ? load_sys(mail).x : *reinterpret_cast<const volatile u32 *>(mail);
            if (s == i) break;
            if (wall_clock64() - t0 > 50000000ull) return;     // 0.5 s: nobody came
        }
        if (mode & 4) dummy[i & 1023] = i;                     // a store to device memory the fence has to write back
        if (mode & 1) store_sys(ack, i);
        else { __threadfence_system(); *reinterpret_cast<volatile u32 *>(ack) = i; }
    }
}
int main() {
    u32 *mail, *ack, *dummy;
    CK(hipHostMalloc((void **)&mail, 64, hipHostMallocDefault));
    CK(hipHostMalloc((void **)&ack, 64, hipHostMallocDefault));
    CK(hipMalloc((void **)&dummy, 4096));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const u32 rounds = 2000;
    const char *names[8] = {"volatile poll, fence_system + store", "volatile poll, sc0 sc1 store", "b128 sc0 sc1 poll, fence_system + store", "b128 sc0 sc1 poll, sc0 sc1 store",
                            "volatile poll, fence_system + store, device store before", "volatile poll, sc0 sc1 store, device store before",
                            "b128 poll, fence_system + store, device store before", "b128 poll, sc0 sc1 store, device store before"};
    for (int mode = 0; mode < 8; ++mode) {
        *(volatile u32 *)mail = 0; *(volatile u32 *)ack = 0;
        hipLaunchKernelGGL(k_pingpong, dim3(1), dim3(64), 0, st, mail, ack, rounds, mode, dummy);
        auto t0 = std::chrono::steady_clock::now();
        for (u32 i = 1; i <= rounds; ++i) {
            *(volatile u32 *)mail = i;
            auto w0 = std::chrono::steady_clock::now();
            while (*(volatile u32 *)ack != i)
                if (std::chrono::steady_clock::now() - w0 > std::chrono::seconds(1)) { printf("mode %d: no answer at round %u\n", mode, i); return 1; }
        }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
        CK(hipStreamSynchronize(st));
        printf("%-58s %6.2f us per round trip\n", names[mode], us);
    }
    return 0;
}
