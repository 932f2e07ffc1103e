// Which CUs does a CU-masked HIP stream (hipExtStreamCreateWithCUMask) reach on MI355X (8 XCDs x 32 CUs)?  Every workgroup of a
// probe kernel records the XCC / SE / SH / CU it ran on; the program prints, per mask, the number of distinct CUs per XCD.
// Round 4, VERDICT r3 item 4 (tail kernels on CUs of their own).   hipcc --offload-arch=gfx950 -O2 cumask_probe.hip -o cumask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_where(uint32_t *out, int spin) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < (uint64_t)spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

static int run(const char *name, hipStream_t st, uint32_t *d, std::vector<uint32_t> &h, int wgs) {
    hipLaunchKernelGGL(k_where, dim3(wgs), dim3(256), 0, st, d, 20000);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h.data(), d, wgs * 8, hipMemcpyDeviceToHost));
    std::map<int, std::set<uint32_t>> per;
    for (int i = 0; i < wgs; ++i) {
        const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per[(int)xcc].insert((se << 8) | (sh << 4) | cu);
    }
    int total = 0;
    printf("%-44s", name);
    for (auto &kv : per) { printf(" x%d:%2zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  = %d CUs\n", total);
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("%s, %d CUs\n", p.gcnArchName, p.multiProcessorCount);
    const int wgs = 8192;
    uint32_t *d;
    CK(hipMalloc(&d, wgs * 8));
    std::vector<uint32_t> h(2 * wgs);
    hipStream_t plain;
    CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
    if (run("unmasked stream", plain, d, h, wgs)) return 1;
    struct M { const char *name; std::vector<uint32_t> w; };
    std::vector<M> masks = {
        {"8 words all ones", std::vector<uint32_t>(8, 0xffffffffu)},
        {"word 0 = ffffffff, rest 0", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}},
        {"word 0 = 000000ff, rest 0", {0x000000ffu, 0, 0, 0, 0, 0, 0, 0}},
        {"every word 0000000f", std::vector<uint32_t>(8, 0x0000000fu)},
        {"every word fffffff0", std::vector<uint32_t>(8, 0xfffffff0u)},
        {"every word 11111111", std::vector<uint32_t>(8, 0x11111111u)},
        {"every word eeeeeeee", std::vector<uint32_t>(8, 0xeeeeeeeeu)},
        {"1 word ffffffff (size 1)", {0xffffffffu}},
        {"1 word 0000ffff (size 1)", {0x0000ffffu}},
        {"1 word 000000ff (size 1)", {0x000000ffu}},
        {"2 words 000000ff 00000000", {0x000000ffu, 0}},
    };
    for (auto &m : masks) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)m.w.size(), m.w.data());
        if (e != hipSuccess) { printf("%-44s hipExtStreamCreateWithCUMask: %s\n", m.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        if (run(m.name, s, d, h, wgs)) return 1;
        CK(hipStreamDestroy(s));
    }
    return 0;
}
