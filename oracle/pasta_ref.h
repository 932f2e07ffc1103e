/* CPU restatement of the Pasta-curve MSM hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (libreef_msm.so) neither links nor calls it.
 *
 * PARITY UNPINNED BY THE REFERENCE: see oracle/pasta_oracle.py header.  The arithmetic
 * restated here lives in crates that are not under /root/reference (nova-snark @
 * sga001/Nova, fil_pasta_curves 0.5.2, pasta-msm); Reef only calls it
 * (src/backend/commitment.rs:187,350,361,371,383,422,430; src/backend/framework.rs:668,695).
 *
 * Layouts (fil_pasta_curves built with `repr-c`, Cargo.toml:14):
 *   field element : 4 x u64 little-endian limbs, Montgomery form, R = 2^256
 *   affine point  : {x, y}      64 B, identity = (0, 0)
 *   jacobian point: {x, y, z}   96 B, identity has z = 0
 * `curve`: 0 = Pallas (coords mod p, scalars mod q), 1 = Vesta (swapped).
 */
#ifndef PASTA_REF_H
#define PASTA_REF_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PASTA_PALLAS = 0, PASTA_VESTA = 1 };

/* sum_i scalars[i]*bases[i] by per-point double-and-add (definition of the MSM). */
void pasta_ref_msm_naive(int curve, const uint64_t *bases_affine, const uint64_t *scalars,
                         size_t n, int scalars_are_mont, uint64_t *out_jacobian);

/* Bucket method restating halo2-style cpu_best_multiexp (window c = 1|3|ceil(ln n),
 * 256/c+1 segments, running-sum bucket reduction, one chunk per thread). */
void pasta_ref_msm_pippenger(int curve, const uint64_t *bases_affine, const uint64_t *scalars,
                             size_t n, int scalars_are_mont, int threads, uint64_t *out_jacobian);

/* Window-parallel bucket method on a persistent thread pool -- the shape of pasta-msm's own CPU path [recalled]: one
 * window size for the whole input (pasta_ref_window_plan), Booth-recoded signed digits, (window, point-slice) tiles dealt
 * out to the pool, running-sum reduction per tile, Horner over the windows.  This is the timed cpu_baseline of bench.py. */
void pasta_ref_msm_pippenger_windows(int curve, const uint64_t *bases_affine, const uint64_t *scalars,
                                     size_t n, int scalars_are_mont, int threads, uint64_t *out_jacobian);
unsigned pasta_ref_window_plan(size_t n, int threads, unsigned *slices_out);
int pasta_ref_pool_size(void);   /* helper threads the pool holds (pthread_create may grant fewer than asked for) */

/* Jacobian (96 B) -> affine (64 B, Montgomery; identity -> (0,0)) and 32-byte compressed form
 * (LE canonical x, y parity in bit 255; identity = zeros). */
void pasta_ref_to_affine(int curve, const uint64_t *jac, size_t n, uint64_t *out_affine);
void pasta_ref_compress(int curve, const uint64_t *jac, size_t n, uint8_t *out32);

/* k * P for one affine point; scalar canonical (not Montgomery) 4xu64. */
void pasta_ref_scalar_mul(int curve, const uint64_t *base_affine, const uint64_t *k_canonical,
                          uint64_t *out_jacobian);

/* Bases in arithmetic progression B_i = (k0 + i*d)*G (affine Montgomery, 64 B each). */
void pasta_ref_gen_bases_ap(int curve, uint64_t k0, uint64_t d, size_t n, uint64_t *out_affine);

/* Seeded scalars (SplitMix64 stream, 4 words per scalar, top bit cleared, one conditional
 * subtract), written in Montgomery form if to_mont. kind: 0 uniform, 1 witness-like
 * (70% {0,1}, 20% < 2^16, 10% uniform), 2 small (value < small_bound). */
void pasta_ref_gen_scalars(int curve, uint64_t seed, int kind, uint64_t small_bound, size_t n,
                           int to_mont, uint64_t *out);

/* Field helpers for unit tests: out = a*b (Montgomery), a+b, a-b, a^-1, to/from Montgomery.
 * field: 0 = Fp, 1 = Fq. */
void pasta_ref_fmul(int field, const uint64_t *a, const uint64_t *b, uint64_t *out);
void pasta_ref_fadd(int field, const uint64_t *a, const uint64_t *b, uint64_t *out);
void pasta_ref_fsub(int field, const uint64_t *a, const uint64_t *b, uint64_t *out);
void pasta_ref_finv(int field, const uint64_t *a, uint64_t *out);
void pasta_ref_to_mont(int field, const uint64_t *a, uint64_t *out);
void pasta_ref_from_mont(int field, const uint64_t *a, uint64_t *out);

/* IPA generator fold (restates CommitmentGens::fold of nova-snark's pedersen provider as
 * used by ipa_pc, reached from src/backend/framework.rs:695): out_i = w1*L_i + w2*R_i,
 * i < half, L = gens[0..half], R = gens[half..2*half]; w canonical; output affine Montgomery. */
void pasta_ref_fold(int curve, const uint64_t *gens_affine, size_t half, const uint64_t *w1,
                    const uint64_t *w2, uint64_t *out_affine);

void pasta_ref_fold_mt(int curve, const uint64_t *gens_affine, size_t half, const uint64_t *w1,
                       const uint64_t *w2, int threads, uint64_t *out_affine);

/* Hyrax-style row commitments (restates HyraxPC::commit, src/backend/commitment.rs:187):
 * out_r = sum_j Z[r*row_len + j]*G_j (+ blinds[r]*H if blinds != NULL). Jacobian out. */
void pasta_ref_row_msm(int curve, const uint64_t *bases_affine, const uint64_t *h_affine,
                       const uint64_t *scalars, const uint64_t *blinds, size_t rows,
                       size_t row_len, int scalars_are_mont, int threads, uint64_t *out_jacobian);

/* Row N2 (sum-check of nlookup witness generation, r1cs_helper.rs:441-506): tables of canonical
 * integers are moved to Montgomery form once, then each round returns (xsq, x, con) and folds both
 * tables with the caller's challenge. */
void pasta_ref_sc_to_mont(int field, uint64_t *table, size_t n);
void pasta_ref_sc_from_mont(int field, uint64_t *table, size_t n);
void pasta_ref_sc_round(int field, uint64_t *T, uint64_t *E, size_t pow, const uint64_t *r_canon, uint64_t *out3);

#ifdef __cplusplus
}
#endif
#endif
