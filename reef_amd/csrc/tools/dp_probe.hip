// DP-FMA against v_mad_u64_u32 for the multiply part of a 255-bit field product on gfx950.
//
// On this chip v_fma_f64 issues at about the rate of v_mad_u64_u32 (profiles/r01_ubench_instruction_rates.txt), so a
// field product on the FP64 pipe with 51/52-bit limbs needs only 25 limb products where nine 29-bit limbs need 81.
// What the count of limb products hides: a DP product of two 51-bit limbs is 102 bits wide and an FMA keeps 53 of them,
// so every limb product is split into a high and a low half and -- unlike v_mad_u64_u32, whose 64-bit accumulator
// takes 15 partial products without a carry -- the split leaves no free accumulation:
//      h  = fma(a, b, C1)        C1 = 3 * 2^102: ulp 2^51, h = C1 + round(ab / 2^51) * 2^51        (round to nearest)
//      t  = C1 - h               = -hi, exact
//      l  = fma(a, b, t)         = ab - hi = lo in [-2^50, 2^50], exact
//      lo_col[k]     += l        |sum of 5| <= 5 * 2^50 < 2^53: exact
//      hi_col[k + 1] -= t        multiples of 2^51: exact while the sum stays below 2^104, i.e. for four terms -- the
//                                fifth term of the middle column goes to an accumulator of its own
// = FIVE instructions per limb product, 125 for the 25 of a product, before any modular reduction (a chained
// h' = fma(a', b', h) saves the subtraction only while C1 + hi + hi' stays below 2^104: two products, and then costs
// the difference h - h' back; summing the bit patterns of h and l with 64-bit integer additions -- Emmart's form --
// is five as well).  The integer form is 81 v_mad_u64_u32 for the same 17 column sums.
//
// This probe (1) checks on the device that both forms give the same 510-bit products for random 255-bit operands and
// (2) times both multiply parts at full occupancy.  Build: hipcc --offload-arch=gfx950 -O3 dp_probe.hip -o dp_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned __int128 u128;
typedef __int128 i128;
typedef uint64_t u64;
typedef uint32_t u32;

static constexpr u32 MASK29 = (1u << 29) - 1;
static constexpr u64 MASK51 = ((u64)1 << 51) - 1;

// ---- multiply part, integer: 17 column sums of nine 29-bit limbs (81 v_mad_u64_u32) ----
__device__ __forceinline__ void int_columns(const u32 (&a)[9], const u32 (&b)[9], u64 (&t)[17]) {
#pragma unroll
    for (int k = 0; k < 17; ++k) t[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) t[i + j] += (u64)a[j] * b[i];
}

// ---- multiply part, FP64: column sums of five 51-bit limbs, high and low halves (125 DP instructions) ----
// lo[k] = sum of the low halves of column k; hi[k] = sum of the high halves of column k - 1 (multiples of 2^51); hi[10] = the
// fifth high half of the middle column, which belongs to hi[5] (five terms of up to 2^102 do not fit 53 bits above 2^51)
__device__ __forceinline__ void dp_columns(const double (&a)[5], const double (&b)[5], double (&lo)[9], double (&hi)[11]) {
    const double C1 = 0x1.8p+103;   // 3 * 2^102
#pragma unroll
    for (int k = 0; k < 9; ++k) lo[k] = 0.0;
#pragma unroll
    for (int k = 0; k < 11; ++k) hi[k] = 0.0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const double h = __builtin_fma(a[j], b[i], C1);
            const double t = C1 - h;
            const double l = __builtin_fma(a[j], b[i], t);
            lo[i + j] += l;
            hi[(i == 4 && j == 0) ? 10 : i + j + 1] -= t;
        }
}

// ---- verification: both forms as 512-bit integers ----
__device__ void pack_int(const u64 (&t)[17], u64 (&out)[8]) {
    u128 acc = 0;
    u32 limbs[18];
    for (int k = 0; k < 17; ++k) { acc += t[k]; limbs[k] = (u32)acc & MASK29; acc >>= 29; }
    limbs[17] = (u32)acc;
    for (int w = 0; w < 8; ++w) out[w] = 0;
    for (int k = 0; k < 18; ++k) {
        const int bit = 29 * k, w = bit >> 6, s = bit & 63;
        if (w < 8) out[w] |= (u64)limbs[k] << s;
        if (s > 35 && w + 1 < 8) out[w + 1] |= (u64)limbs[k] >> (64 - s);
    }
}
__device__ void pack_dp(const double (&lo)[9], const double (&hi)[11], u64 (&out)[8]) {
    i128 acc = 0;
    u64 limbs[11];
    for (int k = 0; k < 10; ++k) {
        const long long l = k < 9 ? (long long)lo[k] : 0;
        const long long h = (long long)(hi[k] * 0x1p-51);   // exact: hi[k] is a multiple of 2^51
        acc += l;
        acc += h;
        if (k == 5) acc += (long long)(hi[10] * 0x1p-51);
        limbs[k] = (u64)acc & MASK51;
        acc >>= 51;
    }
    limbs[10] = (u64)acc;
    for (int w = 0; w < 8; ++w) out[w] = 0;
    for (int k = 0; k < 11; ++k) {
        const int bit = 51 * k, w = bit >> 6, s = bit & 63;
        if (w < 8) out[w] |= limbs[k] << s;
        if (s > 13 && w + 1 < 8) out[w + 1] |= limbs[k] >> (64 - s);
    }
}
__device__ __forceinline__ u64 splitmix(u64 &s) {
    u64 z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ void split_operand(const u64 (&x)[4], u32 (&l29)[9], double (&l51)[5]) {
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, w = bit >> 6, s = bit & 63;
        u64 v = x[w] >> s;
        if (s > 35 && w + 1 < 4) v |= x[w + 1] << (64 - s);
        l29[i] = (u32)v & MASK29;
    }
    for (int i = 0; i < 5; ++i) {
        const int bit = 51 * i, w = bit >> 6, s = bit & 63;
        u64 v = x[w] >> s;
        if (s > 13 && w + 1 < 4) v |= x[w + 1] << (64 - s);
        l51[i] = (double)(v & MASK51);
    }
}
__global__ void __launch_bounds__(256) k_verify(u32 cases_per_thread, u64 seed, unsigned long long *mismatch, unsigned long long *checksum) {
    u64 s = seed + (u64)(blockIdx.x * blockDim.x + threadIdx.x) * 0x1000003ull;
    u64 bad = 0, sum = 0;
    for (u32 c = 0; c < cases_per_thread; ++c) {
        u64 x[4], y[4];
        for (int i = 0; i < 4; ++i) { x[i] = splitmix(s); y[i] = splitmix(s); }
        if ((c & 7) == 0) { for (int i = 0; i < 4; ++i) x[i] = ~0ull; }          // edge: all limbs at their maximum
        if ((c & 15) == 1) { for (int i = 0; i < 4; ++i) y[i] = ~0ull; }
        x[3] &= 0x7fffffffffffffffull;                                             // 255 bits
        y[3] &= 0x7fffffffffffffffull;
        u32 a29[9], b29[9];
        double a51[5], b51[5];
        split_operand(x, a29, a51);
        split_operand(y, b29, b51);
        u64 t[17];
        int_columns(a29, b29, t);
        double lo[9], hi[11];
        dp_columns(a51, b51, lo, hi);
        u64 pi[8], pd[8];
        pack_int(t, pi);
        pack_dp(lo, hi, pd);
        for (int w = 0; w < 8; ++w) { bad += pi[w] != pd[w]; sum += pi[w]; }
    }
    if (bad) atomicAdd(mismatch, (unsigned long long)bad);
    atomicAdd(checksum, (unsigned long long)sum);
}

// ---- timing: ITERS multiply parts per thread, one operand limb fed back from the result each time ----
constexpr int ITERS = 512;
__global__ void __launch_bounds__(256) k_time_int(u32 *out, u32 seed) {
    u32 a[9], b[9];
    for (int i = 0; i < 9; ++i) { a[i] = (seed * (i + 3) + threadIdx.x * 977u) & MASK29; b[i] = (seed * (i + 11) + threadIdx.x * 131u) & MASK29; }
    u32 acc = 0;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 9; ++i) { asm volatile("" : "+v"(a[i])); asm volatile("" : "+v"(b[i])); }   // opaque: nothing of the product is loop-invariant
        u64 t[17];
        int_columns(a, b, t);
        u64 x = 0;
#pragma unroll
        for (int k = 0; k < 17; ++k) x ^= t[k];                   // every column is live: 32 v_xor_b32 (half-cost instructions) on top of the 81 MADs
        a[0] = (u32)x & MASK29;
        acc += (u32)(x >> 32);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_time_dp(u32 *out, u32 seed) {
    double a[5], b[5];
    for (int i = 0; i < 5; ++i) { a[i] = (double)(((u64)seed * (i + 3) * 0x9E3779B97ull + threadIdx.x * 977ull) & MASK51); b[i] = (double)(((u64)seed * (i + 11) * 0xC2B2AE3D27ull + threadIdx.x * 131ull) & MASK51); }
    double acc = 0;
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 5; ++i) { asm volatile("" : "+v"(a[i])); asm volatile("" : "+v"(b[i])); }   // opaque: nothing of the product is loop-invariant
        double lo[9], hi[11];
        dp_columns(a, b, lo, hi);
        u64 x = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) x ^= (u64)__double_as_longlong(lo[k]);   // every column is live: the same kind of v_xor_b32 overhead (36)
#pragma unroll
        for (int k = 1; k < 11; ++k) x ^= (u64)__double_as_longlong(hi[k]);
        a[0] = __builtin_fabs(lo[0]);                             // an integer below 2^51 again (|lo| <= 2^50)
        acc += (double)(u32)x;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(long long)(acc * 0x1p-40);
}

typedef void (*kern_t)(u32 *, u32);
static double time_kernel(kern_t fn, int blocks_per_cu, u32 *out) {
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
    return (double)blocks * 256 * ITERS / (ms * 1e-3);   // multiply parts per second chip-wide
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d\n", prop.gcnArchName, prop.multiProcessorCount);
    unsigned long long *d, h[2] = {0, 0};
    CHECK(hipMalloc(&d, 16));
    CHECK(hipMemset(d, 0, 16));
    const u32 cases = 64, blocks = 1024;
    hipLaunchKernelGGL(k_verify, dim3(blocks), dim3(256), 0, 0, cases, 0x5eefull, d, d + 1);
    CHECK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    printf("verify: %llu products of random 255-bit operands (1 in 8 with an all-ones operand), DP columns == integer columns: %s (mismatching words: %llu, checksum %016llx)\n",
           (unsigned long long)cases * blocks * 256, h[0] == 0 ? "yes" : "NO", h[0], h[1]);
    u32 *out;
    CHECK(hipMalloc(&out, 256 * 16 * 256 * sizeof(u32)));
    for (int bpc : {4, 8}) {
        const double ri = time_kernel(k_time_int, bpc, out), rd = time_kernel(k_time_dp, bpc, out);
        printf("blocks/CU=%d  multiply part of a 255-bit product, chip-wide:  9 x 29-bit limbs on v_mad_u64_u32: %.1f G/s   5 x 51-bit limbs on v_fma_f64: %.1f G/s   DP / integer = %.2f\n",
               bpc, ri * 1e-9, rd * 1e-9, rd / ri);
    }
    return h[0] == 0 ? 0 : 1;
}
