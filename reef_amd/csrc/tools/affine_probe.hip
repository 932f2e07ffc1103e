// affine_probe -- the batched-affine bucket accumulation BUILT once, as an experiment with a kill criterion (VERDICT r4 item 8;
// DESIGN.md 5c had rejected it on estimates).  Same work as k_accum0 on the 2^20 plan: 131072 threads (two waves per SIMD) each sum
// L = 128 table points gathered by index, 16.7 M additions in all.
//   A  XYZZ mixed additions, one after the other (what k_accum0 does: 8M + 2S per addition, nothing leaves the registers).
//   B  affine additions in a pairwise tree, seven levels; the inversions of a level's m/2 additions shared by Montgomery's trick inside
//      the THREAD (prefix products out to memory and back; one Fermat inversion per thread and level).
//   C  the same with the inversion FOR FREE (the level's product is used in place of its inverse: wrong points, right instruction and
//      byte counts) -- the bound for ANY way of sharing inversions further (across lanes, waves, workgroups): whatever that sharing costs
//      comes on top of C.  5M + 1S per addition.
// B is checked against A point for point (both normalised to affine on the device).  Kill criterion: C must beat A by 1.3x for the idea to
// be worth a kernel; otherwise one measured "tried" row replaces the estimate.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I..
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ec.h"
using namespace reef;
constexpr int C = 0;                       // Pallas
constexpr u32 LOGN = 20, N = 1u << LOGN, L = 128, T = 131072;

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ u32 pick(u32 tid, u32 j) {          // the table entry thread tid adds at position j (a stand-in for sorted entries)
    u32 z = tid * 0x9E3779B1u + j * 0x85EBCA77u + 0x165667B1u;
    z ^= z >> 15; z *= 0x2C1B3C6Du; z ^= z >> 12; z *= 0x297A2D39u; z ^= z >> 15;
    return (z & (N - 2)) | (j & 1u);                          // even positions take even entries, odd ones odd: the two operands of a pair are never the same point
}
__device__ __forceinline__ affine load_pt(const affine256 *p) {
    affine a;
    a.x = fe_from_table(p->x);
    a.y = fe_from_table(p->y);
    return a;
}
__device__ __forceinline__ void store_pt(affine256 *p, const fe &x, const fe &y) {
    p->x = fe_to_table<C>(x);
    p->y = fe_to_table<C>(y);
}

// table[i] = k_i * G with k_i a 62-bit hash of i, affine, key-table form.  (NOT an arithmetic progression: with (i + 1) * G two subtrees of a
// thread's pairwise tree sum to the same point now and then -- equal sums of small indices, 5 of 131072 threads -- and the affine formula has no
// doubling case: a real kernel needs that branch, this probe leaves it out, in the tree's favour.)
__global__ void __launch_bounds__(256) k_table(affine256 *tab) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    affine g;
    g.x = fe_canon<C>(fe_neg<C, 2>(fe_one<C>()));
    g.y = fe_canon<C>(fe_dbl<C>(fe_one<C>()));
    u64 h = (u64)(i + 1) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    u32 kw[8] = {(u32)h | 1u, (u32)(h >> 32) & 0x3fffffffu, 0, 0, 0, 0, 0, 0};
    const xyzz p = xyzz_scalar_mul<C>(g, kw, 61);
    const fe izz = fe_inv<C>(p.zz), izzz = fe_inv<C>(p.zzz);
    store_pt(tab + i, fe_mul<C>(p.x, izz), fe_mul<C>(p.y, izzz));
}

// A: the XYZZ chain; the sums are normalised by a kernel of their own (not timed: k_accum0 leaves XYZZ sums too)
__global__ void __launch_bounds__(256) k_normalise(const xyzz_mem *__restrict__ in, affine256 *__restrict__ out) {
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    const xyzz acc = xyzz_from_mem(in[tid]);
    const fe izz = fe_inv<C>(acc.zz), izzz = fe_inv<C>(acc.zzz);
    store_pt(out + tid, fe_mul<C>(acc.x, izz), fe_mul<C>(acc.y, izzz));
}
__global__ void __launch_bounds__(256) k_xyzz(const affine256 *__restrict__ tab, xyzz_mem *__restrict__ out) {
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    xyzz acc = xyzz_identity();
    affine nxt = load_pt(tab + pick(tid, 0));
    for (u32 j = 0; j < L; ++j) {
        const affine cur = nxt;
        if (j + 1 < L) nxt = load_pt(tab + pick(tid, j + 1));     // the next gather is in flight under this addition, as in k_accum0
        acc = xyzz_madd_flag<C>(acc, j == 0, cur);
    }
    out[tid] = xyzz_to_mem(acc);
}

// B / C: the pairwise tree.  work[j * T + tid]: the thread's level arrays (coalesced over the lanes), pre[j * T + tid]: prefix products.
template <bool FREE_INVERSION>
__global__ void __launch_bounds__(256) k_affine_tree(const affine256 *__restrict__ tab, affine256 *__restrict__ work, fe256 *__restrict__ pre,
                                                     affine256 *__restrict__ out) {
    const u32 tid = blockIdx.x * blockDim.x + threadIdx.x;
    u32 base_in = 0;                                   // level >= 2: where this level's inputs start in work[]
    for (u32 m = L; m >= 2; m >>= 1) {
        const u32 half = m >> 1;
        const bool first = m == L;
        const u32 base_out = first ? 0 : base_in + m;
        auto in_pt = [&](u32 j) -> affine { return first ? load_pt(tab + pick(tid, j)) : load_pt(work + (size_t)(base_in + j) * T + tid); };
        // forward: prefix products of the x differences
        fe run = fe_one<C>();
        for (u32 j = 0; j < half; ++j) {
            const affine a = in_pt(2 * j), b = in_pt(2 * j + 1);
            pre[(size_t)j * T + tid] = fe_to_table<C>(run);
            run = fe_mul<C>(run, fe_sub<C, 2>(b.x, a.x));
        }
        fe inv = FREE_INVERSION ? run : fe_inv<C>(run);
        // backward: the additions
        for (u32 jj = half; jj-- > 0;) {
            const affine a = in_pt(2 * jj), b = in_pt(2 * jj + 1);
            const fe dx = fe_sub<C, 2>(b.x, a.x);
            const fe dxinv = fe_mul<C>(inv, fe_from_table(pre[(size_t)jj * T + tid]));
            inv = fe_mul<C>(inv, dx);
            const fe lam = fe_mul<C>(fe_sub<C, 2>(b.y, a.y), dxinv);
            const fe x3 = fe_sub<C, 2>(fe_sub<C, 2>(fe_sqr<C>(lam), a.x), b.x);          // < 5.1
            const fe y3 = fe_sub<C, 2>(fe_mul<C>(lam, fe_sub<C, 8>(a.x, x3)), a.y);
            affine256 *dst = half == 1 ? out + tid : work + (size_t)(base_out + jj) * T + tid;
            store_pt(dst, x3, y3);
        }
        base_in = base_out;
    }
}

int main() {
    affine256 *tab, *work, *oa, *ob;
    xyzz_mem *xa;
    fe256 *pre;
    HIPCK(hipMalloc(&xa, (size_t)T * sizeof(xyzz_mem)));
    HIPCK(hipMalloc(&tab, (size_t)N * sizeof(affine256)));
    HIPCK(hipMalloc(&work, (size_t)L * T * sizeof(affine256)));
    HIPCK(hipMalloc(&pre, (size_t)(L / 2) * T * sizeof(fe256)));
    HIPCK(hipMalloc(&oa, (size_t)T * sizeof(affine256)));
    HIPCK(hipMalloc(&ob, (size_t)T * sizeof(affine256)));
    hipLaunchKernelGGL(k_table, dim3(N / 256), dim3(256), 0, 0, tab);
    HIPCK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    HIPCK(hipEventCreate(&e0));
    HIPCK(hipEventCreate(&e1));
    auto timed = [&](const char *name, auto launch, double products_per_add, double bytes_per_add) {
        launch();
        HIPCK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int r = 0; r < 5; ++r) {
            HIPCK(hipEventRecord(e0));
            launch();
            HIPCK(hipEventRecord(e1));
            HIPCK(hipEventSynchronize(e1));
            float ms;
            HIPCK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double adds = (double)T * (L - 1);
        printf("%-58s %8.3f ms  %6.2f G additions/s  (%.1f field products and %.0f bytes of memory traffic per addition by construction)\n", name, best, adds / best / 1e6,
               products_per_add, bytes_per_add);
        return best;
    };
    const float ta = timed("A  XYZZ chain (k_accum0's form)", [&] { hipLaunchKernelGGL(k_xyzz, dim3(T / 256), dim3(256), 0, 0, tab, xa); }, 9.6, 64.0);
    hipLaunchKernelGGL(k_normalise, dim3(T / 256), dim3(256), 0, 0, xa, oa);
    // bytes per addition of the tree: both operands read twice (2 x 128), prefix product out and back (64), the sum written (64) = 384 (level one reads the table instead)
    const float tb = timed("B  affine tree, inversion per thread and level (Fermat)", [&] { hipLaunchKernelGGL((k_affine_tree<false>), dim3(T / 256), dim3(256), 0, 0, tab, work, pre, ob); },
                           5.8 + 7 * 290.0 / 127, 384.0);
    // B's points are the right ones
    std::vector<affine256> ha(T), hb(T);
    HIPCK(hipMemcpy(ha.data(), oa, (size_t)T * sizeof(affine256), hipMemcpyDeviceToHost));
    HIPCK(hipMemcpy(hb.data(), ob, (size_t)T * sizeof(affine256), hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (u32 i = 0; i < T; ++i) {
        const bool d = memcmp(&ha[i], &hb[i], sizeof(affine256)) != 0;
        if (d && bad < 8) {
            std::vector<u32> pk(L);
            for (u32 j = 0; j < L; ++j) { u32 z = i * 0x9E3779B1u + j * 0x85EBCA77u + 0x165667B1u; z ^= z >> 15; z *= 0x2C1B3C6Du; z ^= z >> 12; z *= 0x297A2D39u; z ^= z >> 15; pk[j] = (z & (N - 2)) | (j & 1u); }
            int dups = 0;
            for (u32 j = 0; j < L; ++j) for (u32 k = j + 1; k < L; ++k) if (pk[j] == pk[k]) { ++dups; fprintf(stderr, "   thread %u: positions %u and %u pick entry %u\n", i, j, k, pk[j]); }
            fprintf(stderr, "   thread %u differs; %d duplicate picks\n", i, dups);
        }
        bad += d;
    }
    printf("   B against A: %zu of %u sums differ%s\n", bad, T, bad ? "  <-- MISMATCH" : " (bit-exact)");
    const float tc = timed("C  affine tree, inversion for free (bound for any sharing)", [&] { hipLaunchKernelGGL((k_affine_tree<true>), dim3(T / 256), dim3(256), 0, 0, tab, work, pre, ob); }, 5.8, 384.0);
    printf("   C / A = %.2fx (kill criterion: the tree with FREE inversions must be 1.3x faster than the XYZZ chain); B / A = %.2fx\n", ta / tc, ta / tb);
    return bad ? 1 : 0;
}
