// The window combine of a key WITHOUT pre-shifted tables, on a host core: sum_g 2^(c*g) * S_g over the G window sums an MSM leaves in the
// landing zone -- a chain of c*(G-1) ~ 255 dependent doublings, ~1 ms on one GPU wave (k_final).  Rounds 2-5 ran the chain through ec.h's
// own formulas compiled for the host: nine 29-bit limbs shaped for v_mad_u64_u32, which hipcc's host pass turns into 0.4 ms of scalar code
// (g++ makes 0.11 ms of the same source; profiles/r06_plain_key_timeline.txt shows the 0.43 ms between the last kernel and the result).
// This file is the same group law (XYZZ: dbl-2008-s-1, add-2008-s, a = 0) on four 64-bit limbs with 128-bit products -- the shape a CPU
// wants -- in Montgomery form R = 2^256, which is also the ABI's: the result needs no conversion.  Host only; tests/test_host_math.py
// compares it with ec.h's chain and with the big-integer oracle, special cases included.
#pragma once
#include <stdint.h>
#include <string.h>

#include "ec.h"

namespace reef {
namespace hostcombine {

typedef unsigned __int128 u128;
struct f4 {
    u64 l[4];
};
template <int F> struct Field {
    f4 p;          // the modulus
    u64 inv;       // -p^-1 mod 2^64
    f4 c251;       // 2^251: mont(v, c251) = v * 2^-5      (v = x * 2^261, the engine's internal form -> x * 2^256)
    f4 c507;       // 2^507 mod p: the same for the bits of v above 2^256
};

static inline bool geq(const f4 &a, const f4 &b) {
    for (int i = 3; i >= 0; --i)
        if (a.l[i] != b.l[i]) return a.l[i] > b.l[i];
    return true;
}
static inline f4 sub_raw(const f4 &a, const f4 &b, u64 *borrow) {
    f4 r;
    u64 br = 0;
    for (int i = 0; i < 4; ++i) {
        const u128 d = (u128)a.l[i] - b.l[i] - br;
        r.l[i] = (u64)d;
        br = (u64)(d >> 64) & 1u;
    }
    *borrow = br;
    return r;
}
static inline f4 add_raw(const f4 &a, const f4 &b) {       // no carry out for operands below 2^255
    f4 r;
    u64 c = 0;
    for (int i = 0; i < 4; ++i) {
        const u128 s = (u128)a.l[i] + b.l[i] + c;
        r.l[i] = (u64)s;
        c = (u64)(s >> 64);
    }
    return r;
}
static inline bool is_zero(const f4 &a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
static inline bool equal(const f4 &a, const f4 &b) { return memcmp(a.l, b.l, sizeof a.l) == 0; }

template <int F> static inline f4 fadd(const Field<F> &fd, const f4 &a, const f4 &b) {   // a, b < p < 2^255
    const f4 s = add_raw(a, b);
    u64 br;
    return geq(s, fd.p) ? sub_raw(s, fd.p, &br) : s;
}
template <int F> static inline f4 fsub(const Field<F> &fd, const f4 &a, const f4 &b) {
    u64 br;
    const f4 d = sub_raw(a, b, &br);
    return br ? add_raw(d, fd.p) : d;
}
// a * b * 2^-256 mod p, fully reduced, for a < 2^256 and b < p (the result of the interleaved reduction is below b + p < 2p)
template <int F> static inline f4 fmul(const Field<F> &fd, const f4 &a, const f4 &b) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        u128 c = 0;
        for (int j = 0; j < 4; ++j) {
            c += (u128)a.l[i] * b.l[j] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (u64)c;
        t[5] = (u64)(c >> 64);
        const u64 m = t[0] * fd.inv;
        c = ((u128)m * fd.p.l[0] + t[0]) >> 64;
        for (int j = 1; j < 4; ++j) {
            c += (u128)m * fd.p.l[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (u64)c;
        t[4] = t[5] + (u64)(c >> 64);
    }
    f4 r = {{t[0], t[1], t[2], t[3]}};
    u64 br;
    if (t[4] || geq(r, fd.p)) r = sub_raw(r, fd.p, &br);
    return r;
}
template <int F> static inline f4 fsqr(const Field<F> &fd, const f4 &a) { return fmul(fd, a, a); }
template <int F> static inline f4 fdbl(const Field<F> &fd, const f4 &a) { return fadd(fd, a, a); }

template <int F> static Field<F> make_field() {
    Field<F> fd;
    u64 t[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 9; ++i) {                           // the modulus from its 29-bit limbs (field_consts.h): one source of truth
        const unsigned sh = 29u * (unsigned)i, w = sh / 64, b = sh % 64;
        const u128 v = (u128)FC<F>::MOD[i] << b;
        u128 s = (u128)t[w] + (u64)v;
        t[w] = (u64)s;
        s = (u128)t[w + 1] + (u64)(v >> 64) + (u64)(s >> 64);
        t[w + 1] = (u64)s;
    }
    for (int i = 0; i < 4; ++i) fd.p.l[i] = t[i];
    u64 x = 1;                                              // Newton: x <- x * (2 - p0 * x) doubles the correct low bits
    for (int i = 0; i < 6; ++i) x *= 2 - fd.p.l[0] * x;
    fd.inv = (u64)0 - x;
    f4 v = {{1, 0, 0, 0}};
    for (int k = 1; k <= 507; ++k) {
        v = fdbl(fd, v);
        if (k == 251) fd.c251 = v;
    }
    fd.c507 = v;
    return fd;
}
template <int F> static const Field<F> &field() {
    static const Field<F> fd = make_field<F>();
    return fd;
}

// nine raw limbs of the engine (29-bit positions, limbs up to 32 bits: un-normalised sums are allowed; value = x * 2^261 mod p, any
// representative below 2^264) -> x * 2^256 mod p, canonical
template <int F> static inline f4 from_engine(const Field<F> &fd, const u32 *w) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 9; ++i) {
        const unsigned sh = 29u * (unsigned)i, k = sh / 64, b = sh % 64;
        const u128 v = (u128)w[i] << b;
        u128 s = (u128)t[k] + (u64)v;
        t[k] = (u64)s;
        s = (u128)t[k + 1] + (u64)(v >> 64) + (u64)(s >> 64);
        t[k + 1] = (u64)s;
        if (k + 2 < 6) t[k + 2] += (u64)(s >> 64);
    }
    const f4 lo = {{t[0], t[1], t[2], t[3]}}, hi = {{t[4], 0, 0, 0}};
    return fadd(fd, fmul(fd, lo, fd.c251), fmul(fd, hi, fd.c507));
}
static inline fe256 to_abi_words(const f4 &a) {
    fe256 r;
    for (int i = 0; i < 4; ++i) { r.w[2 * i] = (u32)a.l[i]; r.w[2 * i + 1] = (u32)(a.l[i] >> 32); }
    return r;
}

struct pt {                                                 // XYZZ: x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2; the identity has ZZ = 0
    f4 x, y, zz, zzz;
};
template <int F> static inline pt pt_from_engine(const Field<F> &fd, const xyzz_mem &m) {
    pt r;
    r.x = from_engine(fd, m.w); r.y = from_engine(fd, m.w + 9); r.zz = from_engine(fd, m.w + 18); r.zzz = from_engine(fd, m.w + 27);
    return r;
}
// 2P (dbl-2008-s-1, a = 0); the identity maps to itself (ZZ stays 0)
template <int F> static inline pt pt_dbl(const Field<F> &fd, const pt &p) {
    const f4 u = fdbl(fd, p.y), v = fsqr(fd, u), w = fmul(fd, u, v), s = fmul(fd, p.x, v), xx = fsqr(fd, p.x);
    const f4 m = fadd(fd, fdbl(fd, xx), xx);
    pt r;
    r.x = fsub(fd, fsqr(fd, m), fdbl(fd, s));
    r.y = fsub(fd, fmul(fd, m, fsub(fd, s, r.x)), fmul(fd, w, p.y));
    r.zz = fmul(fd, v, p.zz);
    r.zzz = fmul(fd, w, p.zzz);
    return r;
}
// P + Q (add-2008-s), every special case: either operand the identity, Q = P (doubling), Q = -P (falls out as ZZ = 0)
template <int F> static inline pt pt_add(const Field<F> &fd, const pt &a, const pt &b) {
    if (is_zero(a.zz)) return b;
    if (is_zero(b.zz)) return a;
    const f4 u1 = fmul(fd, a.x, b.zz), u2 = fmul(fd, b.x, a.zz), s1 = fmul(fd, a.y, b.zzz), s2 = fmul(fd, b.y, a.zzz);
    const f4 p_ = fsub(fd, u2, u1), r_ = fsub(fd, s2, s1);
    if (is_zero(p_) && is_zero(r_)) return pt_dbl(fd, a);
    const f4 pp = fsqr(fd, p_), ppp = fmul(fd, p_, pp), q = fmul(fd, u1, pp);
    pt r;
    r.x = fsub(fd, fsub(fd, fsqr(fd, r_), ppp), fdbl(fd, q));
    r.y = fsub(fd, fmul(fd, r_, fsub(fd, q, r.x)), fmul(fd, s1, ppp));
    r.zz = fmul(fd, fmul(fd, a.zz, b.zz), pp);
    r.zzz = fmul(fd, fmul(fd, a.zzz, b.zzz), ppp);
    return r;
}

}  // namespace hostcombine

// sum_g 2^(c*g) * gs[g] as the ABI's Jacobian point (X' = X*ZZ, Y' = Y*ZZZ, Z' = ZZ; the identity is (0, 0, 0) like pasta_curves' Ep::identity())
template <int C> static inline jacobian256 host_window_combine(const xyzz_mem *gs, u32 G, u32 c) {
    using namespace hostcombine;
    const Field<C> &fd = field<C>();
    pt acc = pt_from_engine(fd, gs[G - 1]);
    for (int g = (int)G - 2; g >= 0; --g) {
        for (u32 k = 0; k < c; ++k) acc = pt_dbl(fd, acc);
        acc = pt_add(fd, acc, pt_from_engine(fd, gs[g]));
    }
    jacobian256 r;
    memset(&r, 0, sizeof r);
    if (is_zero(acc.zz)) return r;
    r.x = to_abi_words(fmul(fd, acc.x, acc.zz));
    r.y = to_abi_words(fmul(fd, acc.y, acc.zzz));
    r.z = to_abi_words(acc.zz);
    return r;
}

}  // namespace reef
