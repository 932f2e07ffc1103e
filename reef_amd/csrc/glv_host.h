// Host side of the GLV split used by the generator fold: k = k1 + k2*lambda (mod r), |k1|, |k2| < 2^128
// (constants and the bound: tools/gen_glv_consts.py).  Fixed-width integer helpers, no allocation.
#pragma once
#include <stdint.h>
#include <string.h>

#include "field.h"
#include "glv_consts.h"

namespace reef {

struct GlvSplit {
    u32 k1[5], k2[5];   // magnitudes, little-endian 32-bit words (160 bits of room, < 2^128 in practice)
    bool neg1, neg2;
};

namespace glv_detail {
typedef unsigned long long w64;
// out[na + nb] = a * b (unsigned, little-endian 64-bit limbs)
inline void mul(const w64 *a, int na, const w64 *b, int nb, w64 *out) {
    for (int i = 0; i < na + nb; ++i) out[i] = 0;
    for (int i = 0; i < na; ++i) {
        unsigned __int128 carry = 0;
        for (int j = 0; j < nb; ++j) {
            const unsigned __int128 t = (unsigned __int128)a[i] * b[j] + out[i + j] + carry;
            out[i + j] = (w64)t;
            carry = t >> 64;
        }
        out[i + nb] = (w64)carry;
    }
}
// acc (8 limbs, two's complement) += / -= t (nt limbs, unsigned)
inline void addsub(w64 *acc, const w64 *t, int nt, bool subtract) {
    unsigned __int128 carry = subtract ? 1 : 0;
    for (int i = 0; i < 8; ++i) {
        w64 v = i < nt ? t[i] : 0;
        if (subtract) v = ~v;
        const unsigned __int128 s = (unsigned __int128)acc[i] + v + carry;
        acc[i] = (w64)s;
        carry = s >> 64;
    }
}
// round(k * g / 2^384) for k of 4 limbs, g of 5 limbs: 3 limbs
inline void mul_shift_384(const w64 *k, const w64 *g, w64 *c3) {
    w64 prod[9];
    mul(k, 4, g, 5, prod);
    unsigned __int128 carry = (unsigned __int128)prod[5] + (1ull << 63);   // + 2^383
    carry >>= 64;
    for (int i = 0; i < 3; ++i) {
        const unsigned __int128 s = (unsigned __int128)prod[6 + i] + carry;
        c3[i] = (w64)s;
        carry = s >> 64;
    }
}
inline void to_magnitude(w64 *acc, u32 *out5, bool *neg) {
    *neg = (acc[7] >> 63) != 0;
    if (*neg) {   // two's complement negate
        unsigned __int128 carry = 1;
        for (int i = 0; i < 8; ++i) {
            const unsigned __int128 s = (unsigned __int128)(~acc[i]) + carry;
            acc[i] = (w64)s;
            carry = s >> 64;
        }
    }
    for (int i = 0; i < 5; ++i) out5[i] = (u32)(acc[i / 2] >> (32 * (i & 1)));
}
}  // namespace glv_detail

// k: canonical scalar (< r) as 8 little-endian 32-bit words.  Returns false if a magnitude needs more than 160 bits
// (cannot happen for k < r; the caller then keeps the plain double-and-add).
template <int C> inline bool glv_split(const u32 *k8, GlvSplit *out) {
    using namespace glv_detail;
    typedef GLV<C> K;
    w64 k[4];
    for (int i = 0; i < 4; ++i) k[i] = (w64)k8[2 * i] | ((w64)k8[2 * i + 1] << 32);
    w64 c1[3], c2[3];
    mul_shift_384(k, K::G1, c1);          // |c1|, sign of c1 = sign(b2)
    mul_shift_384(k, K::G2, c2);          // |c2|, sign of c2 = -sign(b1)
    const bool c1_neg = K::B2_NEG, c2_neg = !K::B1_NEG;
    w64 t[5];
    // k1 = k - c1*a1 - c2*a2
    w64 acc1[8] = {k[0], k[1], k[2], k[3], 0, 0, 0, 0};
    mul(c1, 3, K::A1, 2, t);
    addsub(acc1, t, 5, /*subtract=*/!(c1_neg ^ K::A1_NEG));
    mul(c2, 3, K::A2, 2, t);
    addsub(acc1, t, 5, !(c2_neg ^ K::A2_NEG));
    // k2 = -c1*b1 - c2*b2
    w64 acc2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    mul(c1, 3, K::B1, 2, t);
    addsub(acc2, t, 5, !(c1_neg ^ K::B1_NEG));
    mul(c2, 3, K::B2, 2, t);
    addsub(acc2, t, 5, !(c2_neg ^ K::B2_NEG));
    to_magnitude(acc1, out->k1, &out->neg1);
    to_magnitude(acc2, out->k2, &out->neg2);
    for (int i = 3; i < 8; ++i)
        if (acc1[i] | acc2[i]) return false;                     // more than 192 bits: not a valid split
    return (acc1[2] >> 32) == 0 && (acc2[2] >> 32) == 0;
}

}  // namespace reef
