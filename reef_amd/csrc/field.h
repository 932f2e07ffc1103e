// Pasta base/scalar field arithmetic for gfx950 (and, for CPU-side unit tests only, the host).
//
// Fp / Fq of fil_pasta_curves 0.5.2 (Cargo.toml:14 of the reference): 255-bit primes,
// Montgomery form with R = 2^256, stored as 4 x u64 little-endian == 8 x u32 little-endian.
// On the GPU a field element lives in 8 VGPRs; products are formed with v_mad_u64_u32
// (32x32+64 -> 64).  Both moduli have the shape
//      M = 2^254 + (m3:m2:m1:1)        (limbs 4,5,6 = 0, limb 7 = 0x40000000, limb 0 = 1)
// and -M^-1 mod 2^32 = 0xffffffff, so a CIOS reduction round needs only 3 real
// multiplications (by m1, m2, m3), a shift (the 2^254 term) and no multiplication for
// the quotient digit (q = -t0).  88 MADs per Montgomery product instead of 128.
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

struct alignas(16) fe {
    u32 v[8];
};

// F = 0: Fp (Pallas coordinates, Vesta scalars); F = 1: Fq (Vesta coordinates, Pallas scalars).
// q is the CirC modulus Reef hard-codes at src/backend/r1cs_helper.rs:37-38.
template <int F> struct Mod;
template <> struct Mod<0> {
    static constexpr u32 M1 = 0x992d30edu, M2 = 0x094cf91bu, M3 = 0x224698fcu;
    // R mod p, R^2 mod p (verified with big ints)
    static constexpr u32 R1[8] = {0xfffffffdu, 0x34786d38u, 0xe41914adu, 0x992c350bu,
                                  0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
    static constexpr u32 R2[8] = {0x0000000fu, 0x8c78ecb3u, 0x8b0de0e7u, 0xd7d30dbdu,
                                  0xc3c95d18u, 0x7797a99bu, 0x7b9cb714u, 0x096d41afu};
};
template <> struct Mod<1> {
    static constexpr u32 M1 = 0x8c46eb21u, M2 = 0x0994a8ddu, M3 = 0x224698fcu;
    static constexpr u32 R1[8] = {0xfffffffdu, 0x5b2b3e9cu, 0xe3420567u, 0x992c350bu,
                                  0xffffffffu, 0xffffffffu, 0xffffffffu, 0x3fffffffu};
    static constexpr u32 R2[8] = {0x0000000fu, 0xfc9678ffu, 0x891a16e3u, 0x67bb433du,
                                  0x04ccf590u, 0x7fae2310u, 0x7ccfdaa9u, 0x096d41afu};
};
static constexpr u32 MOD_TOP = 0x40000000u;  // limb 7 of both moduli

template <int F> REEF_HD u32 mod_limb(int i) {
    return i == 0 ? 1u : i == 1 ? Mod<F>::M1 : i == 2 ? Mod<F>::M2 : i == 3 ? Mod<F>::M3 : i == 7 ? MOD_TOP : 0u;
}

REEF_HD bool fe_is_zero(const fe &a) {
    return (a.v[0] | a.v[1] | a.v[2] | a.v[3] | a.v[4] | a.v[5] | a.v[6] | a.v[7]) == 0;
}
REEF_HD bool fe_eq(const fe &a, const fe &b) {
    u32 d = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) d |= a.v[i] ^ b.v[i];
    return d == 0;
}
REEF_HD fe fe_zero() {
    fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = 0;
    return r;
}
template <int F> REEF_HD fe fe_one() {  // Montgomery one
    fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = Mod<F>::R1[i];
    return r;
}
REEF_HD fe fe_select(bool c, const fe &a, const fe &b) {  // c ? a : b
    fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}

// r = a - M if a >= M else a      (a < 2^256)
template <int F> REEF_HD fe fe_reduce_once(const fe &a) {
    fe d;
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u64 x = (u64)a.v[i] - mod_limb<F>(i) - borrow;
        d.v[i] = (u32)x;
        borrow = (x >> 32) & 1;
    }
    return fe_select(borrow != 0, a, d);
}

template <int F> REEF_HD fe fe_add(const fe &a, const fe &b) {
    fe s;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u64 x = (u64)a.v[i] + b.v[i] + c;
        s.v[i] = (u32)x;
        c = x >> 32;
    }
    return fe_reduce_once<F>(s);  // a, b < M < 2^255: no carry out of limb 7
}

template <int F> REEF_HD fe fe_sub(const fe &a, const fe &b) {
    fe d, e;
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u64 x = (u64)a.v[i] - b.v[i] - borrow;
        d.v[i] = (u32)x;
        borrow = (x >> 32) & 1;
    }
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u64 x = (u64)d.v[i] + mod_limb<F>(i) + c;
        e.v[i] = (u32)x;
        c = x >> 32;
    }
    return fe_select(borrow != 0, e, d);
}

template <int F> REEF_HD fe fe_neg(const fe &a) {
    fe d;
    u64 borrow = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u64 x = (u64)mod_limb<F>(i) - a.v[i] - borrow;
        d.v[i] = (u32)x;
        borrow = (x >> 32) & 1;
    }
    return fe_select(fe_is_zero(a), a, d);
}

template <int F> REEF_HD fe fe_dbl(const fe &a) { return fe_add<F>(a, a); }

// One CIOS reduction round on the 9-limb accumulator t (t < 2^288): t = (t + q*M) / 2^32
// with q = -t0 mod 2^32.
template <int F> REEF_HD void mont_round(u32 (&t)[9]) {
    const u32 q = 0u - t[0];
    u64 c = (t[0] != 0) ? 1u : 0u;  // t0 + q*1 == 2^32 (or 0)
    u64 x;
    x = (u64)q * Mod<F>::M1 + t[1] + c; t[0] = (u32)x; c = x >> 32;
    x = (u64)q * Mod<F>::M2 + t[2] + c; t[1] = (u32)x; c = x >> 32;
    x = (u64)q * Mod<F>::M3 + t[3] + c; t[2] = (u32)x; c = x >> 32;
    x = (u64)t[4] + c; t[3] = (u32)x; c = x >> 32;
    x = (u64)t[5] + c; t[4] = (u32)x; c = x >> 32;
    x = (u64)t[6] + c; t[5] = (u32)x; c = x >> 32;
    x = ((u64)q << 30) + t[7] + c; t[6] = (u32)x; c = x >> 32;  // q * 2^30 at limb 7
    x = (u64)t[8] + c; t[7] = (u32)x; t[8] = (u32)(x >> 32);
}

// Montgomery product a*b*R^-1 mod M, inputs and output fully reduced.
template <int F> REEF_HD fe fe_mul(const fe &a, const fe &b) {
    u32 t[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            u64 x = (u64)a.v[j] * b.v[i] + t[j] + c;
            t[j] = (u32)x;
            c = x >> 32;
        }
        t[8] += (u32)c;  // t < 2M before the row, so limb 8 cannot overflow
        mont_round<F>(t);
    }
    fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = t[i];
    return fe_reduce_once<F>(r);  // t < 2M < 2^256
}

template <int F> REEF_HD fe fe_sqr(const fe &a) { return fe_mul<F>(a, a); }

// Montgomery -> canonical (multiply by 1): 8 reduction rounds only.
template <int F> REEF_HD fe fe_from_mont(const fe &a) {
    u32 t[9];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = a.v[i];
    t[8] = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) mont_round<F>(t);
    fe r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = t[i];
    return fe_reduce_once<F>(r);
}

template <int F> REEF_HD fe fe_to_mont(const fe &a) {
    fe r2;
#pragma unroll
    for (int i = 0; i < 8; ++i) r2.v[i] = Mod<F>::R2[i];
    return fe_mul<F>(a, r2);
}

// a^(M-2) (Fermat); inv(0) = 0.  M-2 = 2^254 + (m3:m2:m1:1) - 2.
template <int F> REEF_HD fe fe_inv(const fe &a) {
    // exponent limbs
    u32 e[8];
    e[0] = 0xffffffffu;  // 1 - 2 borrows from limb 1
    e[1] = Mod<F>::M1 - 1u;
    e[2] = Mod<F>::M2;
    e[3] = Mod<F>::M3;
    e[4] = e[5] = e[6] = 0;
    e[7] = MOD_TOP;
    fe acc = fe_one<F>();
    for (int i = 254; i >= 0; --i) {
        acc = fe_sqr<F>(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) acc = fe_mul<F>(acc, a);
    }
    return acc;
}

}  // namespace reef
