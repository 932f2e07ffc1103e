// Pallas / Vesta group law for the MSM kernels (y^2 = x^3 + 5, a = 0).
//
// Bucket accumulators use extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ,
// ZZ^3 = ZZZ^2; identity <=> ZZ == 0 mod M): a mixed add costs 8M + 2S and, unlike plain
// Jacobian, the P + (-P) case falls out of the formulas as ZZ = 0 with no branch.  Only
// P + P needs the (rare, divergent) doubling path.
//
// Value bounds (multiples of M, see field.h) kept by every routine here:
//      affine from a key table: x < 1, y < 2 (y < 1 before a conditional negation)
//      XYZZ:                     X < 6, Y < 3.6, ZZ < 1.6, ZZZ < 1.6   (loose: X < 8, Y < 4, Z* < 2)
// The bound of each intermediate is written next to it; tests/test_host_math.py replays
// the formulas on the host with the bounds tracked and asserted (REEF_BOUNDS build).
//
// Data layouts at the C ABI are those of fil_pasta_curves `repr-c` (Cargo.toml:14):
//   EpAffine/EqAffine {x, y} 64 B, identity = (0, 0);   Ep/Eq {x, y, z} 96 B Jacobian.
// C = 0: Pallas (coordinates in Fp);  C = 1: Vesta (coordinates in Fq).  The coordinate
// field index equals the curve index.
#pragma once
#include "field.h"

// wave-wide "any lane" test on the device (skips rarely-needed selects for the whole wave)
#if defined(__HIP_DEVICE_COMPILE__)
#define REEF_ANY(x) __any(x)
#else
#define REEF_ANY(x) (x)
#endif

namespace reef {

// in-register forms (29-bit limbs, internal Montgomery form)
struct affine {
    fe x, y;
};
struct xyzz {
    fe x, y, zz, zzz;
};
// memory forms
struct alignas(16) affine256 {  // ABI points and resident key tables (64 B)
    fe256 x, y;
};
struct alignas(16) jacobian256 {  // ABI results (96 B)
    fe256 x, y, z;
};
struct alignas(16) xyzz_mem {  // engine-internal accumulators: raw limbs, 144 B
    u32 w[36];
};

REEF_HD bool affine_is_inf(const affine &p) { return fe_is_literal_zero(p.x) && fe_is_literal_zero(p.y); }
template <int C> REEF_HD bool xyzz_is_inf(const xyzz &p) { return fe_is_zero<C>(p.zz); }

REEF_HD xyzz xyzz_identity() {
    xyzz r;
    r.x = fe_zero(); r.y = fe_zero(); r.zz = fe_zero(); r.zzz = fe_zero();
    return r;
}
REEF_HD xyzz xyzz_select(bool c, const xyzz &a, const xyzz &b) {
    xyzz r;
    r.x = fe_select(c, a.x, b.x); r.y = fe_select(c, a.y, b.y);
    r.zz = fe_select(c, a.zz, b.zz); r.zzz = fe_select(c, a.zzz, b.zzz);
    return r;
}
template <int C> REEF_HD xyzz xyzz_from_affine(const affine &p) {
    xyzz r;
    const bool inf = affine_is_inf(p);
    r.x = p.x; r.y = p.y;
    r.zz = fe_select(inf, fe_zero(), fe_one<C>());
    r.zzz = r.zz;
    return r;
}
template <int C> REEF_HD affine affine_neg(const affine &p) {  // y < 2 -> y' < 2 (identity stays (0,0))
    affine r;
    r.x = p.x;
    r.y = fe_select(affine_is_inf(p), p.y, fe_neg<C, 2>(p.y));
    return r;
}
template <int C> REEF_HD xyzz xyzz_neg(const xyzz &p) {  // Y < 4 -> Y' < 4
    xyzz r = p;
    r.y = fe_neg<C, 4>(p.y);
    return r;
}

// 2*P for an affine, non-identity P (mdbl-2008-s-1); x < 1, y < 2.
template <int C> REEF_HD xyzz xyzz_dbl_affine(const affine &p) {
    xyzz r;
    const fe u = fe_dbl<C>(p.y);                              // < 4
    const fe v = fe_sqr<C>(u);                                // 16/128      -> < 1.13
    const fe w = fe_mul<C>(u, v);                             // 4.5/128     -> < 1.04
    const fe s = fe_mul<C>(p.x, v);                           //             -> < 1.01
    const fe xx = fe_sqr<C>(p.x);                             //             -> < 1.01
    const fe m = fe_add<C>(fe_dbl<C>(xx), xx);                // < 3.03
    r.x = fe_sub<C, 4>(fe_sqr<C>(m), fe_dbl<C>(s));           // 1.08 + 4    -> < 5.1   (2s < 2.1 < 4)
    r.y = fe_sub<C, 2>(fe_mul<C>(m, fe_sub<C, 8>(s, r.x)),    // 3.03*9.01   -> < 1.22  (x3 < 8)
                       fe_mul<C>(w, p.y));                    // < 1.02 < 2  ; y3 < 3.3
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2*P (dbl-2008-s-1).  Identity maps to identity (ZZ stays = 0 mod M).
template <int C> REEF_HD xyzz xyzz_dbl(const xyzz &p) {
    xyzz r;
    const fe u = fe_dbl<C>(p.y);                              // < 8
    const fe v = fe_sqr<C>(u);                                // 64/128      -> < 1.5
    const fe w = fe_mul<C>(u, v);                             // 12/128      -> < 1.1
    const fe s = fe_mul<C>(p.x, v);                           // 12/128      -> < 1.1
    const fe xx = fe_sqr<C>(p.x);                             // 64/128      -> < 1.5
    const fe m = fe_add<C>(fe_dbl<C>(xx), xx);                // < 4.5
    r.x = fe_sub<C, 4>(fe_sqr<C>(m), fe_dbl<C>(s));           // 1.16 + 4    -> < 5.2   (2s < 2.2 < 4)
    r.y = fe_sub<C, 2>(fe_mul<C>(m, fe_sub<C, 8>(s, r.x)),    // 4.5*9.1     -> < 1.33
                       fe_mul<C>(w, p.y));                    // 1.1*4       -> < 1.04 < 2 ; y3 < 3.4
    r.zz = fe_mul<C>(v, p.zz);                                // < 1.03
    r.zzz = fe_mul<C>(w, p.zzz);                              // < 1.02
    return r;
}

// acc + P, P affine with x < 1, y < 2 (madd-2008-s).  Handles acc = O, P = O, P = -acc
// (falls out as ZZ = 0) and P = acc (doubling branch).
template <int C> REEF_HD xyzz xyzz_madd_flag(const xyzz &a, bool a_known_empty, const affine &p) {
    const bool a_inf = a_known_empty || xyzz_is_inf<C>(a);
    const bool p_inf = affine_is_inf(p);
    const fe pp_ = fe_mul_sub<C, 8>(p.x, a.zz, a.x);          // U2 - X1: 1.02 + 8   -> < 9.02   (X1 < 8)
    const fe rr = fe_mul_sub<C, 4>(p.y, a.zzz, a.y);          // S2 - Y1: 1.04 + 4   -> < 5.04   (Y1 < 4)
    xyzz r;
    if (__builtin_expect(fe_is_zero<C>(pp_) && fe_is_zero<C>(rr) && !a_inf && !p_inf, 0)) {
        r = xyzz_dbl_affine<C>(p);
    } else {
        const fe pp = fe_sqr<C>(pp_);                         // 81.4/128    -> < 1.64
        const fe ppp = fe_mul<C>(pp_, pp);                    // 14.8/128    -> < 1.12
        const fe q = fe_mul<C>(a.x, pp);                      // 13.1/128    -> < 1.11
        fe t;                                                 // PPP + 2Q < 3.34, left un-normalised
#pragma unroll
        for (int i = 0; i < 9; ++i) t.l[i] = ppp.l[i] + 2u * q.l[i];   // limbs < 3 * 2^29 < 2^31 - 4
        REEF_SET_BOUND(t, REEF_GET_BOUND(ppp) + 2.0 * REEF_GET_BOUND(q));
        r.x = fe_sqr_sub<C, 4>(rr, t);                        // 1.2 + 4     -> < 5.2   (t < 4)
        // Y3 = R*(Q - X3) + (-Y1)*PPP: both products share one reduction
        r.y = fe_mul2_add<C>(rr, fe_sub<C, 8>(q, r.x), fe_neg<C, 4>(a.y), ppp);   // (5.04*9.11 + 4*1.12)/128 -> < 1.4
        r.zz = fe_mul<C>(a.zz, pp);                           // < 1.03
        r.zzz = fe_mul<C>(a.zzz, ppp);                        // < 1.02
    }
    if (REEF_ANY(a_inf)) {                                    // O + P = P
        xyzz from_p;
        from_p.x = p.x; from_p.y = p.y; from_p.zz = fe_one<C>(); from_p.zzz = from_p.zz;
        r = xyzz_select(a_inf, from_p, r);
    }
    if (REEF_ANY(p_inf)) r = xyzz_select(p_inf, a, r);        // acc + O = acc
    return r;
}

template <int C> REEF_HD xyzz xyzz_madd(const xyzz &a, const affine &p) { return xyzz_madd_flag<C>(a, false, p); }

// a + b, both XYZZ (add-2008-s), all special cases handled.
template <int C> REEF_HD xyzz xyzz_add(const xyzz &a, const xyzz &b) {
    const bool a_inf = xyzz_is_inf<C>(a);
    const bool b_inf = xyzz_is_inf<C>(b);
    const fe u1 = fe_mul<C>(a.x, b.zz);                       // 16/128      -> < 1.13
    const fe u2 = fe_mul<C>(b.x, a.zz);                       //             -> < 1.13
    const fe s1 = fe_mul<C>(a.y, b.zzz);                      // 8/128       -> < 1.07
    const fe s2 = fe_mul<C>(b.y, a.zzz);                      //             -> < 1.07
    const fe pp_ = fe_sub<C, 2>(u2, u1);                      // < 3.13
    const fe rr = fe_sub<C, 2>(s2, s1);                       // < 3.07
    xyzz r;
    if (__builtin_expect(fe_is_zero<C>(pp_) && fe_is_zero<C>(rr) && !a_inf && !b_inf, 0)) {
        r = xyzz_dbl<C>(a);
    } else {
        const fe pp = fe_sqr<C>(pp_);                         // 9.8/128     -> < 1.08
        const fe ppp = fe_mul<C>(pp_, pp);                    //             -> < 1.03
        const fe q = fe_mul<C>(u1, pp);                       //             -> < 1.01
        const fe t = fe_add<C>(ppp, fe_dbl<C>(q));            // < 3.05
        r.x = fe_sub<C, 4>(fe_sqr<C>(rr), t);                 // 1.08 + 4    -> < 5.1
        r.y = fe_mul2_add<C>(rr, fe_sub<C, 8>(q, r.x), fe_neg<C, 2>(s1), ppp);   // (3.07*9.01 + 3.07*1.03)/128 -> < 1.25
        r.zz = fe_mul<C>(fe_mul<C>(a.zz, b.zz), pp);          // < 1.01
        r.zzz = fe_mul<C>(fe_mul<C>(a.zzz, b.zzz), ppp);      // < 1.01
    }
    r = xyzz_select(a_inf, b, r);
    r = xyzz_select(b_inf, a, r);
    return r;
}

// ---- memory <-> register conversions ---------------------------------------------------
// key table entry (internal form, canonical, packed) -> affine registers
REEF_HD affine affine_from_table(const affine256 &m) {
    affine r;
    r.x = fe_from_table(m.x);
    r.y = fe_from_table(m.y);
    return r;
}
template <int C> REEF_HD affine256 affine_to_table(const affine &p) {
    affine256 m;
    m.x = fe_to_table<C>(p.x);
    m.y = fe_to_table<C>(p.y);
    return m;
}
// ABI point (Montgomery R = 2^256) -> affine registers; (0,0) stays the identity
template <int C> REEF_HD affine affine_from_abi(const affine256 &m) {
    affine r;
    r.x = fe_from_abi<C>(m.x);
    r.y = fe_from_abi<C>(m.y);
    return r;
}
template <int C> REEF_HD affine256 affine_to_abi(const affine &p) {
    affine256 m;
    m.x = fe_to_abi<C>(p.x);
    m.y = fe_to_abi<C>(p.y);
    return m;
}
REEF_HD xyzz xyzz_from_mem(const xyzz_mem &m) {
    xyzz r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r.x.l[i] = m.w[i]; r.y.l[i] = m.w[9 + i]; r.zz.l[i] = m.w[18 + i]; r.zzz.l[i] = m.w[27 + i];
    }
    REEF_SET_BOUND(r.x, 6.0); REEF_SET_BOUND(r.y, 3.6); REEF_SET_BOUND(r.zz, 1.6); REEF_SET_BOUND(r.zzz, 1.6);
    return r;
}
REEF_HD xyzz_mem xyzz_to_mem(const xyzz &p) {
    xyzz_mem m;
#if defined(REEF_BOUNDS)
    if (p.x.bound > 6.0 || p.y.bound > 3.6 || p.zz.bound > 1.6 || p.zzz.bound > 1.6) REEF_BOUND_FAIL("xyzz_to_mem: invariant");
#endif
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        m.w[i] = p.x.l[i]; m.w[9 + i] = p.y.l[i]; m.w[18 + i] = p.zz.l[i]; m.w[27 + i] = p.zzz.l[i];
    }
    return m;
}

// XYZZ -> ABI Jacobian with Z = ZZ:  X' = X*ZZ, Y' = Y*ZZZ, Z' = ZZ  (x = X'/Z'^2, y = Y'/Z'^3).
// Identity is written as (0, 0, 0) like pasta_curves' Ep::identity().
template <int C> REEF_HD jacobian256 xyzz_to_abi_jacobian(const xyzz &p) {
    jacobian256 r;
    const bool inf = xyzz_is_inf<C>(p);
    r.x = fe_to_abi<C>(fe_select(inf, fe_zero(), fe_mul<C>(p.x, p.zz)));    // 16/128
    r.y = fe_to_abi<C>(fe_select(inf, fe_zero(), fe_mul<C>(p.y, p.zzz)));   // 8/128
    r.z = fe_to_abi<C>(fe_select(inf, fe_zero(), p.zz));
    return r;
}
// ABI Jacobian -> XYZZ: ZZ = Z^2, ZZZ = Z^3.
template <int C> REEF_HD xyzz xyzz_from_abi_jacobian(const jacobian256 &m) {
    xyzz r;
    const fe z = fe_from_abi<C>(m.z);
    r.x = fe_from_abi<C>(m.x);
    r.y = fe_from_abi<C>(m.y);
    r.zz = fe_sqr<C>(z);
    r.zzz = fe_mul<C>(r.zz, z);
    return r;
}

// Affine from XYZZ given t = 1/(ZZ*ZZZ):  1/ZZ = t*ZZZ, 1/ZZZ = t*ZZ.  t < 2.
template <int C> REEF_HD affine xyzz_to_affine_with_inv(const xyzz &p, const fe &t) {
    affine r;
    const bool inf = xyzz_is_inf<C>(p);
    const fe izz = fe_mul<C>(t, p.zzz);                       // < 1.04
    const fe izzz = fe_mul<C>(t, p.zz);
    r.x = fe_select(inf, fe_zero(), fe_mul<C>(p.x, izz));     // < 1.07
    r.y = fe_select(inf, fe_zero(), fe_mul<C>(p.y, izzz));
    return r;
}
template <int C> REEF_HD affine xyzz_to_affine(const xyzz &p) {
    return xyzz_to_affine_with_inv<C>(p, fe_inv<C>(fe_mul<C>(p.zz, p.zzz)));
}

// 32-byte pasta encoding of an affine point (GroupEncoding::to_bytes as reached through
// Commitment::compress, src/backend/commitment.rs:195,351,365): canonical little-endian x,
// bit 255 = parity of canonical y; identity = zeros.
template <int C> REEF_HD fe256 affine_compress(const affine &p) {
    fe one = fe_zero();
    one.l[0] = 1;
    REEF_SET_BOUND(one, 1.0);
    fe256 x = fe_pack(fe_canon<C>(fe_mul<C>(p.x, one)));  // mont(x*R', 1) = x
    const fe y = fe_canon<C>(fe_mul<C>(p.y, one));
    x.w[7] |= (y.l[0] & 1u) << 31;
    if (affine_is_inf(p)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x.w[i] = 0;
    }
    return x;
}

// k*P by left-to-right double-and-add, k canonical 256-bit (8 x u32).  Used by key
// generation / blinding kernels where the work is one-off.
template <int C> REEF_HD xyzz xyzz_scalar_mul(const affine &p, const u32 *k, int top_bit) {
    xyzz acc = xyzz_identity();
    for (int i = top_bit; i >= 0; --i) {
        acc = xyzz_dbl<C>(acc);
        if ((k[i >> 5] >> (i & 31)) & 1u) acc = xyzz_madd<C>(acc, p);
    }
    return acc;
}

}  // namespace reef
