// Pasta base/scalar field arithmetic for gfx950 (and, for CPU-side unit tests only, the host).
//
// Fp / Fq of fil_pasta_curves 0.5.2 (Cargo.toml:14 of the reference) are 255-bit primes.  At
// the C ABI an element is 4 x u64 little-endian limbs in Montgomery form with R = 2^256.
//
// Inside the kernels an element is NINE 29-bit limbs in 32-bit registers ("unsaturated"):
// measured on MI355X (profiles/r01_ubench_instruction_rates.txt) v_mad_u64_u32 issues at the
// same rate as any other 3-operand VALU instruction, so the cost of a Montgomery product is
// its instruction count, and 32-bit limbs pay one extra carry instruction per partial
// product.  With 29-bit limbs a 64-bit column accumulator absorbs all 15 partial products of
// a column without a single carry: a product is 81 + 54 MADs and ~60 cheap instructions.
// The internal Montgomery radix is R' = 2^261 (9 x 29); keys are converted once at upload,
// results once at output (field_consts.h: C_IN, C_OUT).  Both moduli are
//      M = 2^254 + delta,  delta < 2^126   =>  limbs (1, M1, M2, M3, M4, 0, 0, 0, 2^22)
// and M = 1 mod 2^29, so the Montgomery quotient digit is q = -t0 mod 2^29 and a reduction
// round costs 6 MADs.
//
// Value bounds.  Limbs are "normalised" when l0..l7 <= 2^29 + 7 (l0 < 2^29 exactly) and the
// top limb holds the rest.  A product needs (A/M)*(B/M) < 128 and returns a value < 2M with
// exact 29-bit limbs; a - b is computed as a + k*M - b with a bias k*M whose limbs are all
// >= 2^31 - 4 (field_consts.h: BIASk), valid for b < k*M.  The bound of every intermediate
// in the group law is stated next to it in ec.h and machine-checked on the host by
// tests/test_host_math.py (REEF_BOUNDS build).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define REEF_HD __host__ __device__ __forceinline__
#else
#define REEF_HD inline
#endif

namespace reef {
typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;
#if defined(__HIPCC__)
// 8- and 16-byte memory accesses are written with HIP's uint2 / uint4 -- structs of scalars -- and the library is built with
// both vectorizer passes off (csrc/Makefile), so they stay dword instructions.  Measured on MI355X that is FASTER than native
// vector types (ext_vector_type: ds/global b128 instructions) wherever it matters here -- the latency-bound tail kernels
// lose 5-15 % to the register tuples of wide accesses, the bandwidth-bound kernels do not care (profiles/r02_vector_types_ab.txt)
// -- and it keeps hipcc's miscompile of wide load results (ec_coop.h, compiler note) out of reach.  REEF_VEC_VARIANT=2
// selects the native vectors for that comparison.
#if defined(REEF_VEC_VARIANT) && REEF_VEC_VARIANT == 2
typedef u32 u32x2 __attribute__((ext_vector_type(2), may_alias));
typedef u32 u32x4 __attribute__((ext_vector_type(4), may_alias));
REEF_HD u32x2 mk_u32x2(u32 a, u32 b) { u32x2 v; v.x = a; v.y = b; return v; }
REEF_HD u32x4 mk_u32x4(u32 a, u32 b, u32 c, u32 d) { u32x4 v; v.x = a; v.y = b; v.z = c; v.w = d; return v; }
#else
typedef uint2 u32x2;
typedef uint4 u32x4;
REEF_HD u32x2 mk_u32x2(u32 a, u32 b) { return make_uint2(a, b); }
REEF_HD u32x4 mk_u32x4(u32 a, u32 b, u32 c, u32 d) { return make_uint4(a, b, c, d); }
#endif
#endif
}  // namespace reef

#include "field_consts.h"

#if defined(REEF_BOUNDS)
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#define REEF_BOUND_FAIL(msg) do { fprintf(stderr, "REEF_BOUNDS violation: %s (%s:%d)\n", msg, __FILE__, __LINE__); abort(); } while (0)
#endif

namespace reef {

static constexpr u32 LIMB_BITS = 29;
static constexpr u32 LIMB_MASK = (1u << LIMB_BITS) - 1u;

// 9 x 29-bit limbs, Montgomery form w.r.t. R' = 2^261.
struct fe {
    u32 l[9];
#if defined(REEF_BOUNDS)
    double bound;  // value < bound * M  (host-side bound tracking only)
#endif
};

// 256-bit packed element as it lives in HBM / at the ABI (8 x u32 little-endian).
struct alignas(16) fe256 {
    u32 w[8];
};

#if defined(REEF_BOUNDS)
#define REEF_SET_BOUND(x, b) ((x).bound = (b))
#define REEF_GET_BOUND(x) ((x).bound)
#else
#define REEF_SET_BOUND(x, b) ((void)0)
#define REEF_GET_BOUND(x) (0.0)
#endif

REEF_HD fe fe_zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = 0;
    REEF_SET_BOUND(r, 0.0);
    return r;
}
template <int F> REEF_HD fe fe_const(const u32 (&c)[9], double bound) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = c[i];
    REEF_SET_BOUND(r, bound);
    (void)bound;
    return r;
}
template <int F> REEF_HD fe fe_one() { return fe_const<F>(FC<F>::ONE, 1.0); }

REEF_HD fe fe_select(bool c, const fe &a, const fe &b) {  // c ? a : b
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = c ? a.l[i] : b.l[i];
#if defined(REEF_BOUNDS)
    r.bound = a.bound > b.bound ? a.bound : b.bound;
#endif
    return r;
}

// Parallel carry: limbs < 2^32 in, limbs l1..l7 <= 2^29 + 7 and l0 < 2^29 out (same value).
REEF_HD fe fe_norm(const fe &a) {
    fe r;
    r.l[0] = a.l[0] & LIMB_MASK;
#pragma unroll
    for (int i = 1; i < 8; ++i) r.l[i] = (a.l[i] & LIMB_MASK) + (a.l[i - 1] >> LIMB_BITS);
    r.l[8] = a.l[8] + (a.l[7] >> LIMB_BITS);
#if defined(REEF_BOUNDS)
    r.bound = a.bound;
    if ((u64)a.l[8] + (a.l[7] >> LIMB_BITS) >= (1ull << 29)) REEF_BOUND_FAIL("top limb overflow in fe_norm");
#endif
    return r;
}

// Sequential carry: every limb < 2^29 exactly (top limb holds the rest).
REEF_HD fe fe_norm_strict(const fe &a) {
    fe r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 t = a.l[i] + c;  // limbs < 2^32 - 8 by the caller's contract
        r.l[i] = t & LIMB_MASK;
        c = t >> LIMB_BITS;
    }
    r.l[8] = a.l[8] + c;
#if defined(REEF_BOUNDS)
    r.bound = a.bound;
#endif
    return r;
}

template <int F> REEF_HD fe fe_add(const fe &a, const fe &b) {
    fe s;
#pragma unroll
    for (int i = 0; i < 9; ++i) s.l[i] = a.l[i] + b.l[i];
    REEF_SET_BOUND(s, REEF_GET_BOUND(a) + REEF_GET_BOUND(b));
    return fe_norm(s);
}
template <int F> REEF_HD fe fe_dbl(const fe &a) { return fe_add<F>(a, a); }

template <int F, int K> REEF_HD const u32 (&fe_bias())[9] {
    static_assert(K == 2 || K == 4 || K == 8 || K == 16 || K == 32, "bias");
    if constexpr (K == 2) return FC<F>::BIAS2;
    else if constexpr (K == 4) return FC<F>::BIAS4;
    else if constexpr (K == 8) return FC<F>::BIAS8;
    else if constexpr (K == 16) return FC<F>::BIAS16;
    else return FC<F>::BIAS32;
}

// a + K*M - b, normalised.  Requires b < K*M (top limb of b <= top limb of the bias) and
// limbs of b < 2^31 - 4, limbs of a <= 2^29 + 7.
template <int F, int K> REEF_HD fe fe_sub(const fe &a, const fe &b) {
    fe d;
#pragma unroll
    for (int i = 0; i < 9; ++i) d.l[i] = a.l[i] + fe_bias<F, K>()[i] - b.l[i];
#if defined(REEF_BOUNDS)
    if (b.bound > (double)K * (1.0 - 1e-5)) REEF_BOUND_FAIL("fe_sub: subtrahend bound exceeds the bias");
    for (int i = 0; i < 9; ++i) {
        i64 t = (i64)a.l[i] + (i64)fe_bias<F, K>()[i] - (i64)b.l[i];
        if (t < 0 || t >= (1ll << 32)) REEF_BOUND_FAIL("fe_sub: limb out of range");
    }
    d.bound = a.bound + K;
#endif
    return fe_norm(d);
}
template <int F, int K> REEF_HD fe fe_neg(const fe &a) { return fe_sub<F, K>(fe_zero(), a); }

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ void mad_acc_1(u64 &acc, u32 a) {  // acc += a (one v_mad_u64_u32 with the inline constant 1)
    u64 sink;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %0" : "+v"(acc), "=&s"(sink) : "v"(a));
}
#else
REEF_HD void mad_acc_1(u64 &acc, u32 a) { acc += a; }
#endif

#if !defined(__HIP_DEVICE_COMPILE__)
// Host forms of the row operations (the gfx950 forms are whole rows of v_mad_u64_u32 in one
// asm statement each: field_mad_gfx950.h).
REEF_HD void mad_row_new(u64 (&t)[10], const u32 (&a)[9], u32 b) {
    for (int j = 0; j < 9; ++j) t[j] = (u64)a[j] * b;
}
REEF_HD void mad_row_acc(u64 (&t)[10], const u32 (&a)[9], u32 b) {
    for (int j = 0; j < 8; ++j) t[j] += (u64)a[j] * b;
    t[8] = (u64)a[8] * b;
}
REEF_HD void mad_row_add(u64 (&t)[10], const u32 (&a)[9], u32 b) {
    for (int j = 0; j < 9; ++j) t[j] += (u64)a[j] * b;
}
REEF_HD void mad_reduce(u64 (&t)[10], u32 q, u32 m1, u32 m2, u32 m3, u32 m4, u32 top) {
    t[0] += q;
    t[1] += (u64)q * m1; t[2] += (u64)q * m2; t[3] += (u64)q * m3; t[4] += (u64)q * m4;
    t[8] += (u64)q * top;
}
template <int I> REEF_HD void sqr_row(u64 (&t)[10], const u32 (&a)[9], const u32 (&a2)[9]) {
    if (I == 0) {
        t[0] = (u64)a[0] * a[0];
        for (int j = 1; j < 9; ++j) t[j] = (u64)a[0] * a2[j];
    } else if (I == 8) {
        t[8] = (u64)a[8] * a[8];
    } else {
        t[I] += (u64)a[I] * a[I];
        for (int j = I + 1; j < 8; ++j) t[j] += (u64)a[I] * a2[j];
        t[8] = (u64)a[I] * a2[8];
    }
}
#else
}  // namespace reef
#include "field_mad_gfx950.h"
namespace reef {
template <int I> __device__ __forceinline__ void sqr_row(u64 (&t)[10], const u32 (&a)[9], const u32 (&a2)[9]) {
    if constexpr (I == 0) sqr_row0(t, a, a2);
    else if constexpr (I == 1) sqr_row1(t, a, a2);
    else if constexpr (I == 2) sqr_row2(t, a, a2);
    else if constexpr (I == 3) sqr_row3(t, a, a2);
    else if constexpr (I == 4) sqr_row4(t, a, a2);
    else if constexpr (I == 5) sqr_row5(t, a, a2);
    else if constexpr (I == 6) sqr_row6(t, a, a2);
    else if constexpr (I == 7) sqr_row7(t, a, a2);
    else sqr_row8(t, a, a2);
}
#endif

// One Montgomery reduction round on the column accumulators t[0..8] (t[0] = current column):
// q = -t0 mod 2^29, t += q*M, the (now zero) low 29 bits of t[0] are dropped into t[1], and the
// accumulators are renamed one column up (free after unrolling).
template <int F> REEF_HD void mont_round(u64 (&t)[10]) {
    const u32 q = (0u - (u32)t[0]) & LIMB_MASK;
    mad_reduce(t, q, FC<F>::M1, FC<F>::M2, FC<F>::M3, FC<F>::M4, 1u << 22);
    t[1] += t[0] >> LIMB_BITS;
#pragma unroll
    for (int j = 0; j < 9; ++j) t[j] = t[j + 1];
}

template <int F> REEF_HD fe mont_finish(const u64 (&t)[10]) {
    fe r;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u64 v = t[i] + c;
        r.l[i] = (u32)v & LIMB_MASK;
        c = v >> LIMB_BITS;
    }
    r.l[8] = (u32)(t[8] + c);
#if defined(REEF_BOUNDS)
    if (t[8] + c >= (1ull << 29)) REEF_BOUND_FAIL("Montgomery product: result top limb overflow");
#endif
    return r;
}

// Montgomery product a*b/R' mod M.  Inputs normalised with (A/M)*(B/M) < 128; output has exact
// 29-bit limbs and value < 2M.  Row i adds a*b_i into columns i..i+8 (t[0..8]) and reduces
// column i.
template <int F> REEF_HD fe fe_mul(const fe &a, const fe &b) {
#if defined(REEF_BOUNDS)
    if (a.bound * b.bound >= 128.0) REEF_BOUND_FAIL("fe_mul: (A/M)(B/M) >= 128");
    for (int i = 0; i < 8; ++i)
        if (a.l[i] > LIMB_MASK + 8 || b.l[i] > LIMB_MASK + 8) REEF_BOUND_FAIL("fe_mul: operand not normalised");
#endif
    u64 t[10];
    t[9] = 0;
    mad_row_new(t, a.l, b.l[0]);
    mont_round<F>(t);
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        mad_row_acc(t, a.l, b.l[i]);
        mont_round<F>(t);
    }
    fe r = mont_finish<F>(t);
    REEF_SET_BOUND(r, 1.0 + REEF_GET_BOUND(a) * REEF_GET_BOUND(b) / 128.0);
    return r;
}

// (a*b + c*d)/R' mod M with one reduction: two product rows per round feed the same column
// accumulators (18 products and 9 reduction terms of < 2^58 per column stay below 2^63).
// Inputs normalised with (A/M)(B/M) + (C/M)(D/M) < 128; output exact 29-bit limbs, value < 2M.
template <int F> REEF_HD fe fe_mul2_add(const fe &a, const fe &b, const fe &c, const fe &d) {
#if defined(REEF_BOUNDS)
    if (a.bound * b.bound + c.bound * d.bound >= 128.0) REEF_BOUND_FAIL("fe_mul2_add: (A/M)(B/M) + (C/M)(D/M) >= 128");
    for (int i = 0; i < 8; ++i)
        if (a.l[i] > LIMB_MASK + 8 || b.l[i] > LIMB_MASK + 8 || c.l[i] > LIMB_MASK + 8 || d.l[i] > LIMB_MASK + 8)
            REEF_BOUND_FAIL("fe_mul2_add: operand not normalised");
#endif
    u64 t[10];
    t[9] = 0;
    mad_row_new(t, a.l, b.l[0]);
    mad_row_add(t, c.l, d.l[0]);
    mont_round<F>(t);
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        mad_row_acc(t, a.l, b.l[i]);
        mad_row_add(t, c.l, d.l[i]);
        mont_round<F>(t);
    }
    fe r = mont_finish<F>(t);
    REEF_SET_BOUND(r, 1.0 + (REEF_GET_BOUND(a) * REEF_GET_BOUND(b) + REEF_GET_BOUND(c) * REEF_GET_BOUND(d)) / 128.0);
    return r;
}

// a*b + K*M - c in one pass: the limbs of (bias - c) are dropped into the column accumulators
// before the final carry, so the subtraction costs 18 instructions and no extra normalisation.
// Requires c < K*M with limbs < 2^31 - 4; result value < (2 + K)*M, exact 29-bit limbs.
template <int F, int K> REEF_HD fe fe_mul_sub(const fe &a, const fe &b, const fe &c) {
#if defined(REEF_BOUNDS)
    if (a.bound * b.bound >= 128.0) REEF_BOUND_FAIL("fe_mul_sub: (A/M)(B/M) >= 128");
    if (c.bound > (double)K * (1.0 - 1e-5)) REEF_BOUND_FAIL("fe_mul_sub: subtrahend bound exceeds the bias");
    for (int i = 0; i < 8; ++i)
        if (a.l[i] > LIMB_MASK + 8 || b.l[i] > LIMB_MASK + 8) REEF_BOUND_FAIL("fe_mul_sub: operand not normalised");
    for (int i = 0; i < 9; ++i)
        if (c.l[i] > fe_bias<F, K>()[i]) REEF_BOUND_FAIL("fe_mul_sub: subtrahend limb exceeds the bias limb");
#endif
    u64 t[10];
    t[9] = 0;
    mad_row_new(t, a.l, b.l[0]);
    mont_round<F>(t);
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        mad_row_acc(t, a.l, b.l[i]);
        mont_round<F>(t);
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) mad_acc_1(t[i], fe_bias<F, K>()[i] - c.l[i]);
    fe r = mont_finish<F>(t);
    REEF_SET_BOUND(r, 1.0 + REEF_GET_BOUND(a) * REEF_GET_BOUND(b) / 128.0 + K);
    return r;
}

// Montgomery square: 45 instead of 81 partial products (a_i*a_j, i < j, taken once with 2*a_j;
// row i holds columns i..i+8 in t[0..8], so a_i*a_j lands in slot j).
template <int F> REEF_HD fe fe_sqr(const fe &a) {
#if defined(REEF_BOUNDS)
    if (a.bound * a.bound >= 128.0) REEF_BOUND_FAIL("fe_sqr: (A/M)^2 >= 128");
    for (int i = 0; i < 8; ++i)
        if (a.l[i] > LIMB_MASK + 8) REEF_BOUND_FAIL("fe_sqr: operand not normalised");
#endif
    u32 a2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) a2[i] = a.l[i] << 1;
    u64 t[10];
    t[9] = 0;
    sqr_row<0>(t, a.l, a2); mont_round<F>(t);
    sqr_row<1>(t, a.l, a2); mont_round<F>(t);
    sqr_row<2>(t, a.l, a2); mont_round<F>(t);
    sqr_row<3>(t, a.l, a2); mont_round<F>(t);
    sqr_row<4>(t, a.l, a2); mont_round<F>(t);
    sqr_row<5>(t, a.l, a2); mont_round<F>(t);
    sqr_row<6>(t, a.l, a2); mont_round<F>(t);
    sqr_row<7>(t, a.l, a2); mont_round<F>(t);
    sqr_row<8>(t, a.l, a2); mont_round<F>(t);
    fe r = mont_finish<F>(t);
    REEF_SET_BOUND(r, 1.0 + REEF_GET_BOUND(a) * REEF_GET_BOUND(a) / 128.0);
    return r;
}

template <int F, int K> REEF_HD fe fe_sqr_sub(const fe &a, const fe &c) {
#if defined(REEF_BOUNDS)
    if (a.bound * a.bound >= 128.0) REEF_BOUND_FAIL("fe_sqr_sub: (A/M)^2 >= 128");
    if (c.bound > (double)K * (1.0 - 1e-5)) REEF_BOUND_FAIL("fe_sqr_sub: subtrahend bound exceeds the bias");
    for (int i = 0; i < 8; ++i)
        if (a.l[i] > LIMB_MASK + 8) REEF_BOUND_FAIL("fe_sqr_sub: operand not normalised");
    for (int i = 0; i < 9; ++i)
        if (c.l[i] > fe_bias<F, K>()[i]) REEF_BOUND_FAIL("fe_sqr_sub: subtrahend limb exceeds the bias limb");
#endif
    u32 a2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) a2[i] = a.l[i] << 1;
    u64 t[10];
    t[9] = 0;
    sqr_row<0>(t, a.l, a2); mont_round<F>(t);
    sqr_row<1>(t, a.l, a2); mont_round<F>(t);
    sqr_row<2>(t, a.l, a2); mont_round<F>(t);
    sqr_row<3>(t, a.l, a2); mont_round<F>(t);
    sqr_row<4>(t, a.l, a2); mont_round<F>(t);
    sqr_row<5>(t, a.l, a2); mont_round<F>(t);
    sqr_row<6>(t, a.l, a2); mont_round<F>(t);
    sqr_row<7>(t, a.l, a2); mont_round<F>(t);
    sqr_row<8>(t, a.l, a2); mont_round<F>(t);
#pragma unroll
    for (int i = 0; i < 9; ++i) mad_acc_1(t[i], fe_bias<F, K>()[i] - c.l[i]);
    fe r = mont_finish<F>(t);
    REEF_SET_BOUND(r, 1.0 + REEF_GET_BOUND(a) * REEF_GET_BOUND(a) / 128.0 + K);
    return r;
}

// Limbs of j*M (strict), j < 64.
template <int F> REEF_HD fe fe_small_multiple(u32 j) {
    fe r;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u64 v = (u64)j * FC<F>::MOD[i] + c;
        r.l[i] = (u32)v & LIMB_MASK;
        c = v >> LIMB_BITS;
    }
    r.l[8] = (u32)((u64)j * FC<F>::MOD[8] + c);
    REEF_SET_BOUND(r, (double)j);
    return r;
}

// a == 0 (mod M) for a normalised a with value < 64*M.  j*M = j mod 2^29, so the low limb
// decides almost always; the exact comparison runs on a rare, divergent path.
template <int F> REEF_HD bool fe_is_zero(const fe &a) {
#if defined(REEF_BOUNDS)
    if (a.bound >= 64.0) REEF_BOUND_FAIL("fe_is_zero: bound >= 64");
#endif
    const u32 j = a.l[0];
    if (__builtin_expect(j >= 64u, 1)) return false;
    const fe s = fe_norm_strict(a);
    const fe m = fe_small_multiple<F>(j);
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) d |= s.l[i] ^ m.l[i];
    return d == 0;
}
// exact all-limbs-zero test (the canonical encoding of the point at infinity)
REEF_HD bool fe_is_literal_zero(const fe &a) {
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) d |= a.l[i];
    return d == 0;
}

// Fully reduced representative: strict limbs, value < M.  Input value < 64*M.
template <int F> REEF_HD fe fe_canon(const fe &a) {
    const fe s = fe_norm_strict(a);
    const u32 q = s.l[8] >> 22;  // floor(value / 2^254) >= floor(value / M)
    fe r;
    i64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        i64 v = (i64)s.l[i] - (i64)((u64)q * FC<F>::MOD[i]) + c;
        r.l[i] = (u32)v & LIMB_MASK;
        c = v >> LIMB_BITS;  // arithmetic
    }
    i64 top = (i64)s.l[8] - (i64)((u64)q * FC<F>::MOD[8]) + c;
    const bool neg = top < 0;  // q was one too large: add M back
    u32 cc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u32 v = r.l[i] + (neg ? FC<F>::MOD[i] : 0u) + cc;
        r.l[i] = v & LIMB_MASK;
        cc = v >> LIMB_BITS;
    }
    r.l[8] = (u32)(top + (neg ? (i64)FC<F>::MOD[8] : 0) + cc);
    REEF_SET_BOUND(r, 1.0);
    return r;
}

// 256-bit packed (value < 2^256) <-> 9 x 29-bit limbs
REEF_HD fe fe_unpack(const fe256 &p) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, w = bit >> 5, s = bit & 31;
        u32 v = p.w[w] >> s;
        if (s > 3 && w + 1 < 8) v |= p.w[w + 1] << (32 - s);
        r.l[i] = (i == 8) ? v : (v & LIMB_MASK);
    }
    REEF_SET_BOUND(r, 4.0);  // any 256-bit value is < 4M; callers that know better overwrite
    return r;
}
REEF_HD fe256 fe_pack(const fe &a) {  // a strict, value < 2^256
    fe256 p;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        const int bit = 32 * w, i = bit / 29, s = bit - 29 * i;
        u32 v = a.l[i] >> s;
        if (i + 1 < 9) v |= a.l[i + 1] << (29 - s);
        if (29 - s + 29 < 32 && i + 2 < 9) v |= a.l[i + 2] << (58 - s);
        p.w[w] = v;
    }
    return p;
}

// ABI form (Montgomery, R = 2^256, canonical) <-> internal form
template <int F> REEF_HD fe fe_from_abi(const fe256 &p) {
    fe x = fe_unpack(p);
    REEF_SET_BOUND(x, 1.0);  // ABI contract: fully reduced
    return fe_mul<F>(x, fe_const<F>(FC<F>::C_IN, 1.0));
}
template <int F> REEF_HD fe256 fe_to_abi(const fe &a) {
    return fe_pack(fe_canon<F>(fe_mul<F>(a, fe_const<F>(FC<F>::C_OUT, 1.0))));
}
// internal form, canonical, packed: the layout of resident key tables in HBM
template <int F> REEF_HD fe256 fe_to_table(const fe &a) { return fe_pack(fe_canon<F>(a)); }
REEF_HD fe fe_from_table(const fe256 &p) {
    fe x = fe_unpack(p);
    REEF_SET_BOUND(x, 1.0);
    return x;
}
// Montgomery (R = 2^256) scalar -> canonical integer, packed (what the digit recoder reads):
// mont(s*2^256, 2^5) = s.
template <int F> REEF_HD fe256 fe_abi_to_integer(const fe256 &p) {
    fe x = fe_unpack(p);
    REEF_SET_BOUND(x, 1.0);
    fe c32 = fe_zero();
    c32.l[0] = 32;
    REEF_SET_BOUND(c32, 1.0);
    return fe_pack(fe_canon<F>(fe_mul<F>(x, c32)));
}
// canonical integer (< M) -> internal form: mont(x, R'^2)
template <int F> REEF_HD fe fe_from_integer(const fe256 &p) {
    fe x = fe_unpack(p);
    REEF_SET_BOUND(x, 1.0);
    return fe_mul<F>(x, fe_const<F>(FC<F>::C_R2, 1.0));
}

// a^(M-2); inv(0) = 0.  M - 2 = 2^254 + delta - 2.
template <int F> REEF_HD fe fe_inv(const fe &a) {
    // exponent as 9 x 29-bit limbs: MOD with limb0 = 1 - 2 -> borrow from limb 1
    u32 e[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) e[i] = FC<F>::MOD[i];
    e[0] = LIMB_MASK;  // 1 - 2 + 2^29
    e[1] -= 1;
    fe acc = fe_one<F>();
    for (int i = 254; i >= 0; --i) {
        acc = fe_sqr<F>(acc);
        const int limb = i / 29, bit = i - 29 * limb;
        if ((e[limb] >> bit) & 1u) acc = fe_mul<F>(acc, a);
    }
    return acc;
}

}  // namespace reef
