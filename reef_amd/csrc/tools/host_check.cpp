// Host build of field.h / ec.h with bound tracking (REEF_BOUNDS): the same formulas the
// kernels run, executed on the CPU so that `pytest -m "not gpu"` can compare them with the
// oracle and machine-check every value-bound comment.  TEST HARNESS ONLY -- the product
// (libreef_msm.so) never links this.
//   g++ -O2 -std=c++17 -DREEF_BOUNDS -shared -fPIC host_check.cpp -o libreef_hostcheck.so
#include <string.h>
#include <vector>

#include "../ec.h"
#include "../glv_host.h"
#include "../host_combine.h"
#include <chrono>

using namespace reef;

template <int F> static void field_op(int op, const fe256 *a, const fe256 *b, fe256 *out) {
    if (op == 4) { *out = fe_to_abi<F>(fe_from_integer<F>(*a)); return; }
    if (op == 5) { *out = fe_abi_to_integer<F>(*a); return; }
    const fe x = fe_from_abi<F>(*a), y = fe_from_abi<F>(*b);
    fe z;
    switch (op) {
        case 0: z = fe_mul<F>(x, y); break;
        case 1: z = fe_add<F>(x, y); break;
        case 2: z = fe_sub<F, 2>(x, y); break;
        case 3: z = fe_inv<F>(x); break;
        case 6: z = fe_neg<F, 2>(x); break;
        default: z = fe_sqr<F>(x); break;
    }
    *out = fe_to_abi<F>(z);
}

template <int C> static void ec_op(int op, const affine256 *pa, const affine256 *qa, const fe256 *k, jacobian256 *out) {
    const affine p = affine_from_abi<C>(*pa), q = affine_from_abi<C>(*qa);
    xyzz r;
    if (op == 0) r = xyzz_madd<C>(xyzz_from_affine<C>(p), q);
    else if (op == 1) {
        xyzz a = xyzz_dbl<C>(xyzz_from_affine<C>(p));
        a = xyzz_madd<C>(a, affine_neg<C>(p));
        xyzz b = xyzz_dbl<C>(xyzz_from_affine<C>(q));
        b = xyzz_madd<C>(b, affine_neg<C>(q));
        r = xyzz_add<C>(a, b);
    } else if (op == 2) r = xyzz_dbl<C>(xyzz_from_affine<C>(p));
    else r = xyzz_scalar_mul<C>(p, k->w, 255);
    *out = xyzz_to_abi_jacobian<C>(r);
}

// Bucket-style accumulation chain: acc = sum_i (+/-) pts[i] with the table round trip the
// kernels use (ABI -> table form -> registers), exercising the steady-state bounds of madd.
template <int C> static void accumulate(const affine256 *pts, const uint8_t *neg, size_t n, jacobian256 *out, affine256 *out_aff, fe256 *out_comp) {
    xyzz acc = xyzz_identity();
    bool empty = true;
    for (size_t i = 0; i < n; ++i) {
        const affine256 tab = affine_to_table<C>(affine_from_abi<C>(pts[i]));
        affine pt = affine_from_table(tab);
        pt.y = fe_select(neg[i] && !affine_is_inf(pt), fe_neg<C, 2>(pt.y), pt.y);
        acc = xyzz_madd_flag<C>(acc, empty, pt);
        empty = false;
        // round trip through the engine-internal memory form
        acc = xyzz_from_mem(xyzz_to_mem(acc));
    }
    *out = xyzz_to_abi_jacobian<C>(acc);
    const affine a = xyzz_to_affine<C>(acc);
    *out_aff = affine_to_abi<C>(a);
    *out_comp = affine_compress<C>(a);
}

// The window combine of a plain key (host_combine.h) against ec.h's own chain.  pts holds two ABI affine points per window; the window's
// sum is their sum (either or both may be the identity (0, 0)), built with the kernels' mixed addition and sent through the engine's memory
// form -- un-normalised limbs and all -- as the landing zone holds it.  Returns the microseconds of (new, old) per combine.
template <int C> static void window_combine(const affine256 *pts, u32 G, u32 c, jacobian256 *out_new, jacobian256 *out_old, double *us_new, double *us_old) {
    std::vector<xyzz_mem> gs(G);
    for (u32 g = 0; g < G; ++g) {
        xyzz s = xyzz_madd_flag<C>(xyzz_identity(), true, affine_from_abi<C>(pts[2 * g]));
        s = xyzz_madd<C>(s, affine_from_abi<C>(pts[2 * g + 1]));
        gs[g] = xyzz_to_mem(s);
    }
    const int reps = 5;
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) *out_new = host_window_combine<C>(gs.data(), G, c);
    *us_new = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) {
        xyzz acc = xyzz_from_mem(gs[G - 1]);
        for (int g = (int)G - 2; g >= 0; --g) {
            for (u32 k = 0; k < c; ++k) acc = xyzz_dbl<C>(acc);
            acc = xyzz_add<C>(acc, xyzz_from_mem(gs[g]));
        }
        *out_old = xyzz_to_abi_jacobian<C>(acc);
    }
    *us_old = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
}

extern "C" {
void host_window_combine_check(int curve, const void *pts, unsigned G, unsigned c, void *out_new, void *out_old, double *us) {
    if (curve == 0) window_combine<0>((const affine256 *)pts, G, c, (jacobian256 *)out_new, (jacobian256 *)out_old, us, us + 1);
    else window_combine<1>((const affine256 *)pts, G, c, (jacobian256 *)out_new, (jacobian256 *)out_old, us, us + 1);
}
void host_field_op(int field, int op, const void *a, const void *b, void *out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        if (field == 0) field_op<0>(op, (const fe256 *)a + i, (const fe256 *)b + i, (fe256 *)out + i);
        else field_op<1>(op, (const fe256 *)a + i, (const fe256 *)b + i, (fe256 *)out + i);
    }
}
void host_ec_op(int curve, int op, const void *p, const void *q, const void *k, void *out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        if (curve == 0) ec_op<0>(op, (const affine256 *)p + i, (const affine256 *)q + i, (const fe256 *)k + i, (jacobian256 *)out + i);
        else ec_op<1>(op, (const affine256 *)p + i, (const affine256 *)q + i, (const fe256 *)k + i, (jacobian256 *)out + i);
    }
}
void host_accumulate(int curve, const void *pts, const uint8_t *neg, size_t n, void *out_jac, void *out_aff, void *out_comp) {
    if (curve == 0) accumulate<0>((const affine256 *)pts, neg, n, (jacobian256 *)out_jac, (affine256 *)out_aff, (fe256 *)out_comp);
    else accumulate<1>((const affine256 *)pts, neg, n, (jacobian256 *)out_jac, (affine256 *)out_aff, (fe256 *)out_comp);
}
// GLV split of a canonical scalar: out = k1[5], k2[5], neg1, neg2 (32-bit words); returns 1 on success
int host_glv_split(int curve, const uint32_t *k8, uint32_t *out12) {
    GlvSplit sp;
    const bool ok = curve == 0 ? glv_split<0>(k8, &sp) : glv_split<1>(k8, &sp);
    memcpy(out12, sp.k1, 20);
    memcpy(out12 + 5, sp.k2, 20);
    out12[10] = sp.neg1;
    out12[11] = sp.neg2;
    return ok ? 1 : 0;
}
// phi(P) = (beta * x, y) through the device/host-shared field code: out = affine ABI point
void host_glv_phi(int curve, const void *p, void *out) {
    if (curve == 0) {
        affine a = affine_from_abi<0>(*(const affine256 *)p);
        a.x = fe_mul<0>(a.x, fe_const<0>(GLV<0>::BETA, 1.0));
        *(affine256 *)out = affine_to_abi<0>(a);
    } else {
        affine a = affine_from_abi<1>(*(const affine256 *)p);
        a.x = fe_mul<1>(a.x, fe_const<1>(GLV<1>::BETA, 1.0));
        *(affine256 *)out = affine_to_abi<1>(a);
    }
}
// raw pack/unpack round trip (8 x u32 <-> 9 x 29-bit limbs)
void host_pack_roundtrip(const void *in, void *out) {
    *(fe256 *)out = fe_pack(fe_norm_strict(fe_unpack(*(const fe256 *)in)));
}
}
