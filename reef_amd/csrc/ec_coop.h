// Group operations spread over the FOUR waves of a 256-thread workgroup (device only).
//
// Why: the tail of an MSM (partial-sum merge, bucket reduction, generator fold) is a chain of dependent
// group operations on a handful of waves.  A lone wave64 already saturates its SIMD's integer pipe
// (one v_mad_u64_u32 per ~4.6 cycles, profiles/r01_lone_wave_issue.txt), so a general XYZZ addition
// -- 14 field products one after the other -- takes 5.6 us whatever the number of active lanes.  The
// products of one addition are not all dependent, though: they form 4 levels (doubling: 3), and a CU
// has four SIMDs.  Here the four waves of a workgroup hold the SAME 64 operand pairs (replicated
// registers, one pair per lane); at each level every wave computes ONE of the level's products for
// all 64 lanes and the results are exchanged through LDS:
//      addition   4 levels (mul, mul, mul, mul2_add)      instead of 12M + 2S in sequence   (3.45 us measured against 7.1 us)
//      doubling   3 levels (sqr, mul, mul2_add)           instead of  6M + 3S               (2.65 us against 3.7 us)
// Every wave leaves with the complete result, so the code around the operation (shuffles, selects,
// loop control) simply runs replicated and wave-uniform decisions (`__any`) agree across the waves.
//
// LDS layout: one "slot" holds one field element per lane, limbs 0..7 as two 16-byte parts and
// limb 8 as a dword, each part contiguous over the lanes: ds_read/write_b128 from 16 consecutive
// lanes touch all 64 banks exactly once (MI355X_MICROARCH.md, LDS table), so no access conflicts.
// Slots are never reused inside one operation and every level ends in a barrier, so one barrier per
// level is enough (a wave cannot reach a slot's next write before all waves passed the three or four
// barriers in between).
//
// Contract: blockDim.x == 256, every thread of the block calls the operation (no early exits), the
// four waves pass identical operands.  Formulas and value bounds are those of ec.h
// (xyzz_add / xyzz_dbl), where they are machine-checked on the host.
#pragma once
#include "ec.h"

#if defined(__HIPCC__)
namespace reef {

static constexpr int COOP_SLOTS = 11;
struct CoopLds {
    u32x4 q[COOP_SLOTS][2][64];
    u32 t[COOP_SLOTS][64];
    u32 flag[64];
};
#if defined(REEF_COOP_JITTER)
// Timing perturbation for race hunting (tools/bisect/README.md): every wave sleeps a role- and site-dependent time around
// each barrier.  A protocol without races gives the same results with it.
__device__ __forceinline__ void coop_jitter(int role, int site) {
    const int k = (role * 5 + site * 3) & 7;
    if (k == 1) __builtin_amdgcn_s_sleep(3);
    else if (k == 2) __builtin_amdgcn_s_sleep(9);
    else if (k == 3) __builtin_amdgcn_s_sleep(20);
    else if (k == 4) __builtin_amdgcn_s_sleep(1);
    else if (k == 5) __builtin_amdgcn_s_sleep(40);
    else if (k == 6) __builtin_amdgcn_s_sleep(14);
    else if (k == 7) __builtin_amdgcn_s_sleep(60);
}
#define COOP_SYNC() do { coop_jitter(threadIdx.x >> 6, __LINE__); __syncthreads(); coop_jitter(threadIdx.x >> 6, __LINE__ + 1); } while (0)
#else
#define COOP_SYNC() __syncthreads()
#endif

// COMPILER NOTE.  hipcc of ROCm 7.2 miscompiles sequences of these operations when values that came out of 8- or 16-byte
// loads (ds_read_b64/b128, written as such or merged from dword loads by the SLP vectorizer or the AMDGPU load-store
// vectorizer) are kept in registers and picked under the role branches of a later operation: in two doublings inlined back
// to back the role that multiplies v*ZZ receives limbs 0..7 of ZZ from registers that were never loaded on its path (the
// eight copies out of the 128-bit load results sit in another role's block; limb 8, a dword load, is copied in the right
// place), so every wave ends with a wrong ZZ.  Deterministic, independent of timing, wrong from -O1 up (tools/bisect/:
// self-checking reproducer, the -opt-bisect-limit history, the ISA excerpt).  Moving every such dword into a register of its
// own with a v_mov_b32 the compiler cannot see through (coop_launder) cures every sequence at every level, with the
// vectorizers on or off; dword loads do too.  The library does both: it is built with both vectorizers off and writes its
// wide accesses as HIP's uint4 structs (field.h), so every access is a dword instruction, and result slots and operands
// loaded from memory still go through coop_launder (coop_get_result here, load_xyzz_coop in msm_kernels.inc; 1-3 %) in case
// a build turns the passes back on.  tests/test_gpu_parity.py::test_group_law runs every pair and triple of operations.
__device__ __forceinline__ void coop_put(CoopLds &L, int slot, int lane, const fe &v) {
    L.q[slot][0][lane] = mk_u32x4(v.l[0], v.l[1], v.l[2], v.l[3]);
    L.q[slot][1][lane] = mk_u32x4(v.l[4], v.l[5], v.l[6], v.l[7]);
    L.t[slot][lane] = v.l[8];
}
// Two readers.  coop_get is for a slot a role reads inside its own branch and consumes at once.  coop_get_result fetches
// the result slots, whose values every wave keeps in registers and picks from under the role branches of the NEXT
// operation: there the dwords of the 16-byte load results are first moved into registers of their own by instructions the
// compiler cannot see through (coop_launder; see the compiler note).  REEF_COOP_WIDE_READS leaves the move out, for the
// reproducer.
__device__ __forceinline__ fe coop_get(const CoopLds &L, int slot, int lane) {
    const u32x4 a = L.q[slot][0][lane], b = L.q[slot][1][lane];
    fe r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w;
    r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = L.t[slot][lane];
    return r;
}
__device__ __forceinline__ u32 coop_launder(u32 v) {
#if defined(REEF_COOP_WIDE_READS)
    return v;
#else
    u32 o;
    asm("v_mov_b32 %0, %1" : "=v"(o) : "v"(v));        // not volatile: free to be scheduled, still opaque
    return o;
#endif
}
__device__ __forceinline__ fe coop_get_result(const CoopLds &L, int slot, int lane) {
    const u32x4 a = L.q[slot][0][lane], b = L.q[slot][1][lane];
    fe r;
    r.l[0] = coop_launder(a.x); r.l[1] = coop_launder(a.y); r.l[2] = coop_launder(a.z); r.l[3] = coop_launder(a.w);
    r.l[4] = coop_launder(b.x); r.l[5] = coop_launder(b.y); r.l[6] = coop_launder(b.z); r.l[7] = coop_launder(b.w);
    r.l[8] = L.t[slot][lane];
    return r;
}
// One group operation by the four waves of the workgroup: mode COOP_ADD: a + b (both XYZZ, every special case of
// xyzz_add in ec.h); mode COOP_DBL: 2*a (xyzz_dbl; b is ignored).  role = threadIdx.x >> 6, lane = threadIdx.x & 63.
//
// Addition and doubling share ONE body (three plain product levels, then a level with a product and a two-product
// form); the mode is a template parameter, the role a wave-uniform scalar, so operand preparation is scalar branches
// around register moves and LDS reads and each level carries a single inlined field product.  Kernels keep to one
// addition site (and one doubling site) per kernel: ~6 field products of code each, where an inlined operation per use
// made the one-wave tail kernels 40-580 KB.  The rare P + P case of an addition (the formulas degenerate to ZZ3 = 0 with
// R = 0) is one more shared doubling.
//
//   level   addition (roles 0..3)                                   doubling (roles 0..3)
//   1       u1 = X1*ZZ2 | u2 = X2*ZZ1 | s1 = Y1*ZZZ2 | s2 = Y2*ZZZ1  v = U^2 (U = 2Y) | xx = X^2 | - | -
//   2       pp = P^2 (P = u2-u1) | zz12 | zzz12 | rr2 = R^2 (R = s2-s1)   w = U*v | s = X*v | mm = M^2 (M = 3xx) | ZZ3 = v*ZZ
//   3       ppp = P*pp | ZZ3 = zz12*pp | q = u1*pp | -                (no third level)
//   4       - | - | ZZZ3 = zzz12*ppp | X3 = rr2-ppp-2q, Y3 = R(q-X3)-s1*ppp     ZZZ3 = w*ZZZ | - | X3 = mm-2s, Y3 = M(s-X3)-w*Y | -
enum { COOP_ADD = 0, COOP_DBL = 1 };
template <int C, int MODE> __device__ __forceinline__ xyzz xyzz_coop_op(const xyzz &a, const xyzz &b, CoopLds &L, int role_, int lane) {
    constexpr bool add = MODE == COOP_ADD;
    const int role = __builtin_amdgcn_readfirstlane(role_);
    fe x, y;
    // ---- level 1
    if constexpr (add) {
        if (role == 0) { x = a.x; y = b.zz; }
        else if (role == 1) { x = b.x; y = a.zz; }
        else if (role == 2) { x = a.y; y = b.zzz; }
        else { x = b.y; y = a.zzz; }
    } else {
        if (role == 0) { x = fe_dbl<C>(a.y); y = x; }         // U < 8
        else { x = a.x; y = a.x; }
    }
    fe l1 = x;
    if (add || role < 2) { l1 = fe_mul<C>(x, y); coop_put(L, role, lane, l1); }
    COOP_SYNC();
    // ---- level 2
    fe keep = x;                                              // add: P (role 0), R (role 3); dbl: U (role 0), M (role 2)
    fe n1 = x;                                                // add, role 3: -s1, made while level 3 runs elsewhere
    if constexpr (add) {
        if (role == 0) { keep = fe_sub<C, 2>(coop_get(L, 1, lane), l1); x = keep; y = keep; }            // P < 3.13
        else if (role == 1) { x = a.zz; y = b.zz; }
        else if (role == 2) { x = a.zzz; y = b.zzz; }
        else { n1 = coop_get(L, 2, lane); keep = fe_sub<C, 2>(l1, n1); x = keep; y = keep; }             // R < 3.07
    } else {
        if (role == 0) { y = l1; }                                                                        // U * v  (x, keep = U)
        else if (role == 1) { x = a.x; y = coop_get(L, 0, lane); }                                        // X * v
        else if (role == 2) { const fe xx = coop_get(L, 1, lane); keep = fe_add<C>(fe_dbl<C>(xx), xx); x = keep; y = keep; }   // M < 4.5
        else { x = coop_get(L, 0, lane); y = a.zz; }                                                      // v * ZZ
    }
    const fe l2 = fe_mul<C>(x, y);
    if constexpr (add) {
        if (role == 0) coop_put(L, 4, lane, l2);              // pp
        if (role == 3) L.flag[lane] = fe_is_zero<C>(keep) ? 1u : 0u;
    } else {
        if (role == 0) coop_put(L, 5, lane, l2);              // w      (the slot of ppp: level 4 reads it as its second factor)
        if (role == 1) coop_put(L, 6, lane, l2);              // s      (the slot of q)
        if (role == 3) coop_put(L, 9, lane, l2);              // ZZ3
    }
    COOP_SYNC();
    // ---- level 3 (addition only)
    if constexpr (add) {
        if (role == 3) {
            n1 = fe_neg<C, 2>(n1);                            // -s1, s1 < 1.07
        } else {
            const fe pp = coop_get(L, 4, lane);               // < 1.08
            if (role == 0) x = keep;
            else if (role == 1) x = l2;
            else x = coop_get(L, 0, lane);                    // u1
            const fe l3 = fe_mul<C>(x, pp);
            coop_put(L, role == 0 ? 5 : role == 2 ? 6 : 9, lane, l3);   // ppp | q | ZZ3
        }
        COOP_SYNC();
    }
    // ---- level 4: one role multiplies ZZZ3, another builds X3 and Y3; slot 5 = ppp | w, slot 6 = q | s
    constexpr int mul_role = add ? 2 : 0, xy_role = add ? 3 : 2;
    if (role == mul_role) {
        fe f;
        if constexpr (add) f = coop_get(L, 5, lane); else f = a.zzz;   // zzz12*ppp | w*ZZZ   (l2 = zzz12 | w)
        coop_put(L, 10, lane, fe_mul<C>(l2, f));
    } else if (role == xy_role) {
        const fe p5 = coop_get(L, 5, lane);                   // ppp < 1.03 | w < 1.1
        const fe p6 = coop_get(L, 6, lane);                   // q < 1.01   | s < 1.1
        fe t;                                                 // ppp + 2q < 3.05 | 2s < 2.2, left un-normalised (limbs < 3 * 2^29 < 2^31 - 4)
#pragma unroll
        for (int i = 0; i < 9; ++i) t.l[i] = (add ? p5.l[i] : 0u) + 2u * p6.l[i];
        const fe x3 = fe_sub<C, 4>(l2, t);                    // rr2 | mm : 1.16 + 4 -> < 5.2
        if constexpr (!add) n1 = fe_neg<C, 4>(a.y);           // -Y (Y < 4)
        coop_put(L, 7, lane, x3);
        coop_put(L, 8, lane, fe_mul2_add<C>(keep, fe_sub<C, 8>(p6, x3), n1, p5));      // R(q - X3) - s1*ppp | M(s - X3) - Y*w: < 1.4
    }
    COOP_SYNC();
    xyzz r;
    r.x = coop_get_result(L, 7, lane);
    r.y = coop_get_result(L, 8, lane);
    r.zz = coop_get_result(L, 9, lane);
    r.zzz = coop_get_result(L, 10, lane);
    if constexpr (add) {
        const bool a_inf = xyzz_is_inf<C>(a), b_inf = xyzz_is_inf<C>(b);
        // P + P: then P = R = 0 and the formulas above give ZZ3 = 0; P + (-P) also gives ZZ3 = 0 but has R != 0.  Every wave
        // holds the same lanes, so the test agrees across the workgroup and the rare doubling is one more shared operation.
        const bool same = L.flag[lane] != 0 && fe_is_zero<C>(r.zz) && !a_inf && !b_inf;
        if (__any(same)) {
            COOP_SYNC();                                  // slots 7..10 and the flag are read everywhere
            const xyzz d = xyzz_coop_op<C, COOP_DBL>(a, a, L, role_, lane);
            r = xyzz_select(same, d, r);
        }
        if (__any(a_inf || b_inf)) {                          // an identity operand: the other one is the sum
            r = xyzz_select(a_inf, b, r);
            r = xyzz_select(b_inf, a, r);
        }
    }
    return r;
}

template <int C> __device__ __forceinline__ xyzz xyzz_add_coop(const xyzz &a, const xyzz &b, CoopLds &L, int role, int lane) {
    return xyzz_coop_op<C, COOP_ADD>(a, b, L, role, lane);
}
template <int C> __device__ __forceinline__ xyzz xyzz_dbl_coop(const xyzz &p, CoopLds &L, int role, int lane) {
    return xyzz_coop_op<C, COOP_DBL>(p, p, L, role, lane);
}

}  // namespace reef
#endif
