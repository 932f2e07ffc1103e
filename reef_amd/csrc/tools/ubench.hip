// Instruction-throughput probes for the integer pipeline of gfx950 (what bounds a 255-bit
// Montgomery product): v_mad_u64_u32, v_mul_lo/hi_u32, 24-bit multiplies, carry adds,
// v_fma_f64, plus LDS and global atomics as used by the bucket sort.
// Build: hipcc --offload-arch=gfx950 -O3 ubench.hip -o ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 2048;
constexpr int UNROLL = 16;  // instructions per loop body per chain

#define LOOP(BODY) for (int it = 0; it < ITERS; ++it) { _Pragma("unroll") for (int u = 0; u < UNROLL / 4; ++u) { BODY; } }
#define TID (blockIdx.x * blockDim.x + threadIdx.x)

// 4 independent chains each
__global__ void __launch_bounds__(256) p_mad64(uint32_t *out, uint32_t seed) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    uint32_t x = seed | 1, y = threadIdx.x | 3;
    LOOP(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y) : "vcc"));
    out[TID] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3);
}
#define U32_PROBE(NAME, YINIT, ASM)                                                        \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed) {            \
        uint32_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;           \
        uint32_t y = threadIdx.x | (YINIT);                                                \
        LOOP(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y) : "vcc")); \
        out[TID] = a0 ^ a1 ^ a2 ^ a3;                                                      \
    }
U32_PROBE(p_mul_lo, 3u, "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4")
U32_PROBE(p_mul_hi, 0x80000003u, "v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4")
U32_PROBE(p_mad_u24, 3u, "v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0")
U32_PROBE(p_mul_hi_u24, 0x800003u, "v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4")
U32_PROBE(p_add_u32, 3u, "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4")
U32_PROBE(p_addc, 0xf0000003u, "v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc")
U32_PROBE(p_cndmask, 3u, "v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc")

__global__ void __launch_bounds__(256) p_fma_f64(uint32_t *out, uint32_t seed) {
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    double y = 1.0000001;
    LOOP(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y)));
    out[TID] = (uint32_t)(a0 + a1 + a2 + a3);
}
__global__ void __launch_bounds__(256) p_fma_f32(uint32_t *out, uint32_t seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float y = 1.0000001f;
    LOOP(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y)));
    out[TID] = (uint32_t)(a0 + a1 + a2 + a3);
}
// mixed: 1 mad64 + 2 carry adds (the shape of a CIOS inner step), twice per statement
__global__ void __launch_bounds__(256) p_mad_mix(uint32_t *out, uint32_t seed) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 + 1;
    uint32_t b0 = seed, b1 = seed + 5, x = seed | 1, y = threadIdx.x | 3;
    LOOP(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_add_co_u32 %2, vcc, %2, %5\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
                      "v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_add_co_u32 %3, vcc, %3, %5\n v_addc_co_u32 %2, vcc, %2, %4, vcc"
                      : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1) : "v"(x), "v"(y) : "vcc"));
    out[TID] = (uint32_t)(a0 ^ a1) ^ b0 ^ b1;
}

U32_PROBE(p_bfi, 3u, "v_bfi_b32 %0, %4, %0, %1\n v_bfi_b32 %1, %4, %1, %2\n v_bfi_b32 %2, %4, %2, %3\n v_bfi_b32 %3, %4, %3, %0")
U32_PROBE(p_xor, 3u, "v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4")
U32_PROBE(p_and_or, 3u, "v_and_or_b32 %0, %0, %4, %1\n v_and_or_b32 %1, %1, %4, %2\n v_and_or_b32 %2, %2, %4, %3\n v_and_or_b32 %3, %3, %4, %0")
U32_PROBE(p_add3, 3u, "v_add3_u32 %0, %0, %4, %1\n v_add3_u32 %1, %1, %4, %2\n v_add3_u32 %2, %2, %4, %3\n v_add3_u32 %3, %3, %4, %0")
U32_PROBE(p_subb, 0xf0000003u, "v_sub_co_u32 %0, vcc, %0, %4\n v_subb_co_u32 %1, vcc, %1, %4, vcc\n v_subb_co_u32 %2, vcc, %2, %4, vcc\n v_subb_co_u32 %3, vcc, %3, %4, vcc")
U32_PROBE(p_cmp_cnd, 3u, "v_cmp_gt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_gt_u32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc")
U32_PROBE(p_alignbit, 3u, "v_alignbit_b32 %0, %0, %1, 29\n v_alignbit_b32 %1, %1, %2, 29\n v_alignbit_b32 %2, %2, %3, 29\n v_alignbit_b32 %3, %3, %0, 29")
U32_PROBE(p_lshr, 3u, "v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3")
U32_PROBE(p_mov, 3u, "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4")

__global__ void __launch_bounds__(256) p_cnd_sgpr(uint32_t *out, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    uint32_t y = threadIdx.x | 3;
    uint64_t m = 0x5555aaaa5555aaaaull ^ seed;
    LOOP(asm volatile("v_cndmask_b32_e64 %0, %0, %4, %5\n v_cndmask_b32_e64 %1, %1, %4, %5\n v_cndmask_b32_e64 %2, %2, %4, %5\n v_cndmask_b32_e64 %3, %3, %4, %5"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y), "s"(m)));
    out[TID] = a0 ^ a1 ^ a2 ^ a3;
}
__global__ void __launch_bounds__(256) p_lshl_add64(uint32_t *out, uint32_t seed) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    uint64_t y = ((uint64_t)threadIdx.x << 33) | 3;
    LOOP(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(y)));
    out[TID] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3);
}
__global__ void __launch_bounds__(256) p_lshr64(uint32_t *out, uint32_t seed) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    LOOP(asm volatile("v_lshrrev_b64 %0, 1, %0\n v_lshrrev_b64 %1, 1, %1\n v_lshrrev_b64 %2, 1, %2\n v_lshrrev_b64 %3, 1, %3"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)));
    out[TID] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3);
}
// mad with carry-out to an SGPR pair consumed by v_addc (the even/odd accumulation shape)
__global__ void __launch_bounds__(256) p_mad_addc(uint32_t *out, uint32_t seed) {
    uint64_t a0 = seed + threadIdx.x, a1 = a0 + 1;
    uint32_t k0 = 0, k1 = 0, x = seed | 0x80000001u, y = threadIdx.x | 0xc0000003u;
    LOOP(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n"
                      "v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_addc_co_u32 %3, vcc, 0, %3, vcc"
                      : "+v"(a0), "+v"(a1), "+v"(k0), "+v"(k1) : "v"(x), "v"(y) : "vcc"));
    out[TID] = (uint32_t)(a0 ^ a1) ^ k0 ^ k1;
}
__global__ void __launch_bounds__(256) p_bpermute(uint32_t *out, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    uint32_t idx = ((threadIdx.x + 5) & 63) * 4;
    LOOP(asm volatile("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)"
                      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(idx)));
    out[TID] = a0 ^ a1 ^ a2 ^ a3;
}

typedef void (*probe_fn)(uint32_t *, uint32_t);

static double run_probe(const char *name, probe_fn fn, int insts_per_iter, int blocks_per_cu, uint32_t *out) {
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(fn, dim3(blocks), dim3(256), 0, 0, out, 12345u);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double lane_ops = (double)blocks * 256 * ITERS * insts_per_iter;
    const double per_s = lane_ops / (ms * 1e-3);
    // cycles per wave-instruction per SIMD assuming 2.4 GHz and 1024 SIMDs
    const double wave_insts = lane_ops / 64.0;
    const double cyc = (ms * 1e-3) * 2.4e9 * 1024.0 / wave_insts;
    printf("%-14s blocks/CU=%d  %8.3f ms  %8.2f Tlane-op/s  ~%5.2f cyc/wave-inst/SIMD (at 2.4 GHz)\n", name, blocks_per_cu, ms,
           per_s * 1e-12, cyc);
    return per_s;
}

// ------------------------------------------------------------ atomics probes ----
__global__ void __launch_bounds__(1024) p_lds_atomic(uint32_t *out, uint32_t nkeys, uint32_t iters, uint32_t hot) {
    extern __shared__ uint32_t h[];
    for (uint32_t k = threadIdx.x; k < nkeys; k += blockDim.x) h[k] = 0;
    __syncthreads();
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        uint32_t k = (s >> 8) % nkeys;
        if (hot && (s & 3)) k = 1;  // 75% of lanes hit one bucket
        acc += atomicAdd(&h[k], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = acc + h[1];
}

__global__ void __launch_bounds__(256) p_global_atomic(uint32_t *table, uint32_t nkeys, uint32_t iters, uint32_t hot, uint32_t *out) {
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 17;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        uint32_t k = (s >> 8) % nkeys;
        if (hot && (s & 3)) k = 1;
        acc += atomicAdd(&table[k], 1u);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// random 64-byte gathers (the bucket accumulation's point fetch)
__global__ void __launch_bounds__(256) p_gather64(const uint4 *table, uint32_t nelem, uint32_t iters, uint32_t *out) {
    uint32_t s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 17;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        const uint4 *p = table + (size_t)((s >> 4) % nelem) * 4;
        uint4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x ^ b.y ^ c.z ^ d.w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    uint32_t *out;
    CHECK(hipMalloc(&out, 256 * 16 * 256 * sizeof(uint32_t)));
    for (int bpc : {2, 8}) {
        run_probe("v_mad_u64_u32", p_mad64, UNROLL, bpc, out);
        run_probe("v_mul_lo_u32", p_mul_lo, UNROLL, bpc, out);
        run_probe("v_mul_hi_u32", p_mul_hi, UNROLL, bpc, out);
        run_probe("v_mad_u32_u24", p_mad_u24, UNROLL, bpc, out);
        run_probe("v_mul_hi_u24", p_mul_hi_u24, UNROLL, bpc, out);
        run_probe("v_add_u32", p_add_u32, UNROLL, bpc, out);
        run_probe("v_addc chain", p_addc, UNROLL, bpc, out);
        run_probe("v_cndmask", p_cndmask, UNROLL, bpc, out);
        run_probe("v_fma_f64", p_fma_f64, UNROLL, bpc, out);
        run_probe("v_fma_f32", p_fma_f32, UNROLL, bpc, out);
        run_probe("mad+2adds x2", p_mad_mix, 6 * (UNROLL / 4), bpc, out);
        run_probe("mad+addc x2", p_mad_addc, 4 * (UNROLL / 4), bpc, out);
        run_probe("v_bfi_b32", p_bfi, UNROLL, bpc, out);
        run_probe("v_xor_b32", p_xor, UNROLL, bpc, out);
        run_probe("v_and_or_b32", p_and_or, UNROLL, bpc, out);
        run_probe("v_add3_u32", p_add3, UNROLL, bpc, out);
        run_probe("v_subb chain", p_subb, UNROLL, bpc, out);
        run_probe("cmp+cndmask", p_cmp_cnd, UNROLL, bpc, out);
        run_probe("cndmask sgpr", p_cnd_sgpr, UNROLL, bpc, out);
        run_probe("v_alignbit", p_alignbit, UNROLL, bpc, out);
        run_probe("v_lshrrev_b32", p_lshr, UNROLL, bpc, out);
        run_probe("v_mov_b32", p_mov, UNROLL, bpc, out);
        run_probe("v_lshl_add_u64", p_lshl_add64, UNROLL, bpc, out);
        run_probe("v_lshrrev_b64", p_lshr64, UNROLL, bpc, out);
        run_probe("ds_bpermute", p_bpermute, UNROLL, bpc, out);
    }
    // LDS atomics: 1024-thread blocks, 32768 counters (128 KiB)
    CHECK(hipFuncSetAttribute((const void *)p_lds_atomic, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (uint32_t hot : {0u, 1u}) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const uint32_t iters = 4096, blocks = 256;
        hipLaunchKernelGGL(p_lds_atomic, dim3(blocks), dim3(1024), 131072, 0, out, 32768u, 16u, hot);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(p_lds_atomic, dim3(blocks), dim3(1024), 131072, 0, out, 32768u, iters, hot);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("LDS atomicAdd (rtn) 32768 keys hot=%u: %.3f ms, %.2f G atomics/s chip-wide\n", hot, ms, (double)blocks * 1024 * iters / ms * 1e-6);
    }
    uint32_t *table;
    CHECK(hipMalloc(&table, (1u << 20) * sizeof(uint32_t)));
    CHECK(hipMemset(table, 0, (1u << 20) * sizeof(uint32_t)));
    for (uint32_t hot : {0u, 1u}) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const uint32_t iters = hot ? 16 : 64, blocks = 2048;
        hipLaunchKernelGGL(p_global_atomic, dim3(blocks), dim3(256), 0, 0, table, 1u << 19, 2u, hot, out);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(p_global_atomic, dim3(blocks), dim3(256), 0, 0, table, 1u << 19, iters, hot, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("global atomicAdd (rtn) 2^19 keys hot=%u: %.3f ms, %.2f G atomics/s\n", hot, ms, (double)blocks * 256 * iters / ms * 1e-6);
    }
    {
        const size_t nelem = (size_t)1 << 24;  // 1 GiB table of 64-byte elements
        uint4 *big;
        CHECK(hipMalloc(&big, nelem * 64));
        CHECK(hipMemset(big, 1, nelem * 64));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const uint32_t iters = 64, blocks = 2048;
        hipLaunchKernelGGL(p_gather64, dim3(blocks), dim3(256), 0, 0, big, (uint32_t)nelem, 2u, out);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(p_gather64, dim3(blocks), dim3(256), 0, 0, big, (uint32_t)nelem, iters, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("random 64 B gathers over 1 GiB: %.3f ms, %.2f G gathers/s = %.1f GB/s\n", ms, (double)blocks * 256 * iters / ms * 1e-6,
               (double)blocks * 256 * iters * 64 / ms * 1e-6);
    }
    return 0;
}
