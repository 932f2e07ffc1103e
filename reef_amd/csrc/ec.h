// Pallas / Vesta group law for the MSM kernels (y^2 = x^3 + 5, a = 0).
//
// Bucket accumulators use extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ,
// ZZ^3 = ZZZ^2; identity <=> ZZ == 0): a mixed add costs 8M + 2S and, unlike plain
// Jacobian, the P + (-P) case falls out of the formulas as ZZ = 0 with no branch.
// Only P + P needs the (rare, divergent) doubling path.
//
// Data layouts at the C ABI are those of fil_pasta_curves `repr-c` (Cargo.toml:14):
//   EpAffine/EqAffine {x, y} 64 B, identity = (0, 0);   Ep/Eq {x, y, z} 96 B Jacobian.
// C = 0: Pallas (coordinates in Fp);  C = 1: Vesta (coordinates in Fq).  The coordinate
// field index equals the curve index.
#pragma once
#include "field.h"

namespace reef {

struct alignas(16) affine {
    fe x, y;
};
struct alignas(16) xyzz {
    fe x, y, zz, zzz;
};
struct alignas(16) jacobian {
    fe x, y, z;
};

REEF_HD bool affine_is_inf(const affine &p) { return fe_is_zero(p.x) && fe_is_zero(p.y); }
REEF_HD bool xyzz_is_inf(const xyzz &p) { return fe_is_zero(p.zz); }

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
    bool inf = affine_is_inf(p);
    r.x = p.x; r.y = p.y;
    r.zz = fe_select(inf, fe_zero(), fe_one<C>());
    r.zzz = r.zz;
    return r;
}
template <int C> REEF_HD affine affine_neg(const affine &p) {
    affine r;
    r.x = p.x; r.y = fe_neg<C>(p.y);
    return r;
}
template <int C> REEF_HD xyzz xyzz_neg(const xyzz &p) {
    xyzz r = p;
    r.y = fe_neg<C>(p.y);
    return r;
}

// 2*P for an affine, non-identity P (mdbl-2008-s-1).
template <int C> REEF_HD xyzz xyzz_dbl_affine(const affine &p) {
    xyzz r;
    fe u = fe_dbl<C>(p.y);
    fe v = fe_sqr<C>(u);
    fe w = fe_mul<C>(u, v);
    fe s = fe_mul<C>(p.x, v);
    fe xx = fe_sqr<C>(p.x);
    fe m = fe_add<C>(fe_dbl<C>(xx), xx);
    r.x = fe_sub<C>(fe_sub<C>(fe_sqr<C>(m), s), s);
    r.y = fe_sub<C>(fe_mul<C>(m, fe_sub<C>(s, r.x)), fe_mul<C>(w, p.y));
    r.zz = v;
    r.zzz = w;
    return r;
}

// 2*P (dbl-2008-s-1).  Identity maps to identity (ZZ stays 0).
template <int C> REEF_HD xyzz xyzz_dbl(const xyzz &p) {
    xyzz r;
    fe u = fe_dbl<C>(p.y);
    fe v = fe_sqr<C>(u);
    fe w = fe_mul<C>(u, v);
    fe s = fe_mul<C>(p.x, v);
    fe xx = fe_sqr<C>(p.x);
    fe m = fe_add<C>(fe_dbl<C>(xx), xx);
    r.x = fe_sub<C>(fe_sub<C>(fe_sqr<C>(m), s), s);
    r.y = fe_sub<C>(fe_mul<C>(m, fe_sub<C>(s, r.x)), fe_mul<C>(w, p.y));
    r.zz = fe_mul<C>(v, p.zz);
    r.zzz = fe_mul<C>(w, p.zzz);
    return r;
}

// acc + P, P affine (madd-2008-s).  Handles acc = O, P = O, P = -acc (falls out) and
// P = acc (doubling branch).
template <int C> REEF_HD xyzz xyzz_madd(const xyzz &a, const affine &p) {
    const bool a_inf = xyzz_is_inf(a);
    const bool p_inf = affine_is_inf(p);
    fe u2 = fe_mul<C>(p.x, a.zz);
    fe s2 = fe_mul<C>(p.y, a.zzz);
    fe pp_ = fe_sub<C>(u2, a.x);
    fe rr = fe_sub<C>(s2, a.y);
    xyzz r;
    if (__builtin_expect(fe_is_zero(pp_) && fe_is_zero(rr) && !a_inf && !p_inf, 0)) {
        r = xyzz_dbl_affine<C>(p);
    } else {
        fe pp = fe_sqr<C>(pp_);
        fe ppp = fe_mul<C>(pp_, pp);
        fe q = fe_mul<C>(a.x, pp);
        r.x = fe_sub<C>(fe_sub<C>(fe_sub<C>(fe_sqr<C>(rr), ppp), q), q);
        r.y = fe_sub<C>(fe_mul<C>(rr, fe_sub<C>(q, r.x)), fe_mul<C>(a.y, ppp));
        r.zz = fe_mul<C>(a.zz, pp);
        r.zzz = fe_mul<C>(a.zzz, ppp);
    }
    xyzz from_p;
    from_p.x = p.x; from_p.y = p.y; from_p.zz = fe_one<C>(); from_p.zzz = from_p.zz;
    r = xyzz_select(a_inf, from_p, r);  // O + P = P
    r = xyzz_select(p_inf, a, r);       // acc + O = acc
    return r;
}

// a + b, both XYZZ (add-2008-s), all special cases handled.
template <int C> REEF_HD xyzz xyzz_add(const xyzz &a, const xyzz &b) {
    const bool a_inf = xyzz_is_inf(a);
    const bool b_inf = xyzz_is_inf(b);
    fe u1 = fe_mul<C>(a.x, b.zz);
    fe u2 = fe_mul<C>(b.x, a.zz);
    fe s1 = fe_mul<C>(a.y, b.zzz);
    fe s2 = fe_mul<C>(b.y, a.zzz);
    fe pp_ = fe_sub<C>(u2, u1);
    fe rr = fe_sub<C>(s2, s1);
    xyzz r;
    if (__builtin_expect(fe_is_zero(pp_) && fe_is_zero(rr) && !a_inf && !b_inf, 0)) {
        r = xyzz_dbl<C>(a);
    } else {
        fe pp = fe_sqr<C>(pp_);
        fe ppp = fe_mul<C>(pp_, pp);
        fe q = fe_mul<C>(u1, pp);
        r.x = fe_sub<C>(fe_sub<C>(fe_sub<C>(fe_sqr<C>(rr), ppp), q), q);
        r.y = fe_sub<C>(fe_mul<C>(rr, fe_sub<C>(q, r.x)), fe_mul<C>(s1, ppp));
        r.zz = fe_mul<C>(fe_mul<C>(a.zz, b.zz), pp);
        r.zzz = fe_mul<C>(fe_mul<C>(a.zzz, b.zzz), ppp);
    }
    r = xyzz_select(a_inf, b, r);
    r = xyzz_select(b_inf, a, r);
    return r;
}

// XYZZ -> Jacobian with Z = ZZ:  X' = X*ZZ, Y' = Y*ZZZ, Z' = ZZ  (x = X'/Z'^2, y = Y'/Z'^3).
// Identity is written as (0, 0, 0) like pasta_curves' Ep::identity().
template <int C> REEF_HD jacobian xyzz_to_jacobian(const xyzz &p) {
    jacobian r;
    bool inf = xyzz_is_inf(p);
    r.x = fe_select(inf, fe_zero(), fe_mul<C>(p.x, p.zz));
    r.y = fe_select(inf, fe_zero(), fe_mul<C>(p.y, p.zzz));
    r.z = p.zz;
    return r;
}
// Jacobian -> XYZZ: ZZ = Z^2, ZZZ = Z^3.
template <int C> REEF_HD xyzz jacobian_to_xyzz(const jacobian &p) {
    xyzz r;
    r.x = p.x; r.y = p.y;
    r.zz = fe_sqr<C>(p.z);
    r.zzz = fe_mul<C>(r.zz, p.z);
    return r;
}

// Affine from XYZZ given t = 1/(ZZ*ZZZ):  1/ZZ = t*ZZZ, 1/ZZZ = t*ZZ.
template <int C> REEF_HD affine xyzz_to_affine_with_inv(const xyzz &p, const fe &t) {
    affine r;
    bool inf = xyzz_is_inf(p);
    fe izz = fe_mul<C>(t, p.zzz);
    fe izzz = fe_mul<C>(t, p.zz);
    r.x = fe_select(inf, fe_zero(), fe_mul<C>(p.x, izz));
    r.y = fe_select(inf, fe_zero(), fe_mul<C>(p.y, izzz));
    return r;
}
template <int C> REEF_HD affine xyzz_to_affine(const xyzz &p) {
    return xyzz_to_affine_with_inv<C>(p, fe_inv<C>(fe_mul<C>(p.zz, p.zzz)));
}

// 32-byte pasta encoding of an affine point (GroupEncoding::to_bytes as reached through
// Commitment::compress, src/backend/commitment.rs:195,351,365): canonical little-endian x,
// bit 255 = parity of canonical y; identity = zeros.
template <int C> REEF_HD fe affine_compress(const affine &p) {
    fe x = fe_from_mont<C>(p.x);
    fe y = fe_from_mont<C>(p.y);
    x.v[7] |= (y.v[0] & 1u) << 31;
    return fe_select(affine_is_inf(p), fe_zero(), x);
}

// k*P by left-to-right double-and-add, k canonical 256-bit (8 x u32).  Used by key
// generation / precomputation / fold kernels where every lane shares the control flow
// or the work is one-off.
template <int C> REEF_HD xyzz xyzz_scalar_mul(const affine &p, const u32 *k, int top_bit) {
    xyzz acc = xyzz_identity();
    for (int i = top_bit; i >= 0; --i) {
        acc = xyzz_dbl<C>(acc);
        if ((k[i >> 5] >> (i & 31)) & 1u) acc = xyzz_madd<C>(acc, p);
    }
    return acc;
}

}  // namespace reef
