/* libreef_msm.so -- MI355X (gfx950) backend for the Pasta-curve MSM hot path of eniac/Reef.
 *
 * C ABI only: plain pointers and sizes, no C++ / torch types.  Every entry point names the
 * reference interface it replaces (paths relative to the Reef repository).  Symbols marked
 * [R] belong to crates that are NOT vendored in the reference tree (nova-snark @ sga001/Nova,
 * fil_pasta_curves 0.5.2, pasta-msm; Cargo.toml:12,14) -- their layouts are isolated in this
 * header so they can be corrected in one place.
 *
 * Data layouts (fil_pasta_curves `repr-c`, Cargo.toml:14) [R]:
 *   reef_fe        4 x u64 little-endian limbs, Montgomery form (R = 2^256), fully reduced
 *   reef_affine    {x, y}      64 B, identity = (0, 0)                 (EpAffine / EqAffine)
 *   reef_jacobian  {x, y, z}   96 B, identity has z = 0                (Ep / Eq)
 * Scalars of Pallas are Fq elements, scalars of Vesta are Fp elements (the cycle).
 *
 * Error behaviour: the reference treats every failure as a panic (framework.rs:683,702).  The
 * pasta-msm compatible symbols return void and abort() with a message on any error; the handle
 * API returns a reef_status and records a message retrievable with reef_last_error().  There is
 * no CPU fallback: without a usable gfx950 device every call fails loudly.
 *
 * Threading: all entry points are re-entrant.  A reef_msm_ctx serialises the calls made on it
 * (it owns one HIP stream and one workspace); use one ctx (or clone) per concurrent caller.
 *
 * ABI changelog (reef_abi_version()):
 *   6  round 6: the drop-in symbols build a returning key's resident copy on a builder thread (no call pays for it: reef_key_cache_wait,
 *      reef_key_cache_stats.spares in place of .reserved); REEF_SC_FENCE defaults to the release-ordered ticket; device groups report where a
 *      call's time went (reef_msm_group_enable_timing / _last_timing) and take REEF_SCALARS_FANOUT (reef_msm_group_opts.scalars, was reserved[0]).
 *   5  round 5: device groups (reef_msm_group_*: one MSM split by window or by points, or the rows of a Hyrax commitment dealt out
 *      whole, over several GPUs of ONE process, the partial sums exchanged inside the library: peer copies, host slots, or a
 *      single-process RCCL communicator loaded at run time); reef_merkle_commit_devices (the Merkle tree in blocks over several GPUs); reef_get_device;
 *      reef_key_cache_timing_get; reef_runtime_opts.hw_queues is the only way the library touches GPU_MAX_HW_QUEUES unless
 *      REEF_MSM_HW_QUEUES is exported (the load-time default of rounds 3-4 is gone); the drop-in symbols confirm a returning key's
 *      bytes on the host whatever its size (no key upload on a hit).
 *   4  round 4: reef_runtime_init / reef_abi_version / reef_msm_ctx_attach / reef_msm_multi / reef_key_cache_info added; the drop-in symbols' key cache is one table per process
 *      (clones per calling thread) instead of one cache per thread.
 *   3  round 3: reef_msm_opts.byte_tables = 0 changed meaning from "build the byte tables in the background" to "follow the
 *      process-wide policy: none unless REEF_MSM_WIDE=1" -- callers that pass zeroed opts no longer get the byte-table path
 *      for MSM / IPA unless they ask (1 or 3).  A performance difference only; results are identical.
 *   2  round 2: byte tables, nibble tables, reef_msm_folded, rows N1-N4.
 *   1  round 1.
 */
#ifndef REEF_MSM_H
#define REEF_MSM_H
#define REEF_ABI_VERSION 6

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t l[4]; } reef_fe;
typedef struct { reef_fe x, y; } reef_affine;
typedef struct { reef_fe x, y, z; } reef_jacobian;

enum { REEF_PALLAS = 0, REEF_VESTA = 1 };
enum { REEF_HOST = 0, REEF_DEVICE = 1 }; /* where a buffer lives */

typedef enum {
    REEF_OK = 0,
    REEF_ERR_ARG = 1,    /* bad argument (null pointer, size mismatch, unknown curve) */
    REEF_ERR_HIP = 2,    /* HIP runtime error (message in reef_last_error) */
    REEF_ERR_NO_GPU = 3, /* no gfx950 device visible */
    REEF_ERR_OOM = 4
} reef_status;

/* ---------------------------------------------------------------------------------------------
 * (1) pasta-msm drop-in symbols [R].
 * Replaces: `extern "C" mult_pippenger_pallas/vesta` declared by the pasta-msm crate and reached
 * from nova-snark's `Group::vartime_multiscalar_mul` (provider/pasta.rs) for every commitment:
 * src/backend/framework.rs:668 (RecursiveSNARK::prove_step), :695 (CompressedSNARK::prove),
 * src/backend/commitment.rs:187,350,361,371,383,422,430.
 * Stateless: nothing is retained; bases are uploaded per call.  `is_mont` = scalars are in
 * Montgomery form (what the Rust wrapper passes).  abort()s on error.
 * ------------------------------------------------------------------------------------------- */
/* (A key that keeps coming back is nominated by a non-cryptographic hash of 64 sampled points (no pass over all of its bytes), CONFIRMED byte for byte
 * against a retained HOST copy while the GPU already works on the nominated key (helper threads share the comparison of
 * keys above 4 MiB), and served from a resident pre-shifted copy as soon as one exists; a hash collision
 * therefore costs time, never a wrong result, and nothing the caller can observe is retained.  The resident copy is built OFF the
 * caller's thread: the call that brings a key's second appearance is served like the first, keeps a copy of the key's bytes and
 * returns; one builder thread of the process uploads the copy, builds the tables and prepares a ready context for the first call
 * that finds the key published (typically the third or fourth).  A caller never waits for the builder.  The table of resident keys is
 * one per process (at most 16 keys, REEF_MSM_KEY_CACHE_MB of device memory, default 16384, and REEF_MSM_KEY_HOST_MB of host
 * memory, default 4096; a key stays charged while any thread's context is still attached to it): the key is built once, whichever threads call --
 * nova-snark reaches these symbols from the prover thread and from rayon workers, src/backend/framework.rs:110,
 * 668,695 -- and each calling thread serves it through its own stream and workspace on the shared tables.  On an
 * allocation failure the table is emptied and the call is served uncached.  REEF_MSM_KEY_CACHE=0 turns it off.) */
void mult_pippenger_pallas(reef_jacobian *out, const reef_affine *points, size_t npoints,
                           const reef_fe *scalars, bool is_mont);
void mult_pippenger_vesta(reef_jacobian *out, const reef_affine *points, size_t npoints,
                          const reef_fe *scalars, bool is_mont);
/* What the drop-in symbols' process-wide key table holds (tests, diagnostics): entries nominated, keys with a resident copy,
 * device bytes charged to the budget, and counters since the process started -- resident copies built, calls served from
 * one, per-thread contexts made by the calling threads themselves, speculative calls whose bytes turned out to be another key's (served again on the plain path), ready contexts the builder thread prepared.  reef_key_cache_clear drops every entry (threads let go of their clones at their next call). */
typedef struct {
    uint64_t entries, resident_keys, resident_bytes, builds, hits, clones, misspeculated, spares;
} reef_key_cache_stats;
void reef_key_cache_info(reef_key_cache_stats *out);
void reef_key_cache_clear(void);
/* Blocks until the builder thread has nothing left to do (tests and timing harnesses that want a deterministic "the key is resident
 * now"; a prover never needs it). */
void reef_key_cache_wait(void);
/* Where the calls served from a resident key spent their time on the HOST, summed over all calling threads since the last reset
 * (nanoseconds; diagnostics: tools/seam_bench): nominating the key (sampled hash + table lookup), enqueueing the MSM (includes the
 * staging of the caller's pageable scalars), confirming the caller's bytes (memcmp against the retained copy, beside the GPU), and
 * waiting for the result. */
typedef struct {
    uint64_t calls, nominate_ns, enqueue_ns, confirm_ns, wait_ns, reserved[3];
} reef_key_cache_timing;
void reef_key_cache_timing_get(reef_key_cache_timing *out, int reset);

/* ---------------------------------------------------------------------------------------------
 * (2) Resident-key handle API.
 * Replaces: nova-snark `CommitmentGens<G>` [R] as held inside PublicParams / HyraxPC for the
 * life of a proof (src/backend/framework.rs:45,134,297-303; src/backend/commitment.rs:176-186).
 * The bases are uploaded once; with `precompute` the key also stores 2^(c*G*j)-shifted copies so
 * that several Pippenger windows share one bucket set (HBM is 288 GB; a 2^20-point key with
 * 16 tables is 1 GiB).
 * ------------------------------------------------------------------------------------------- */
typedef struct reef_msm_ctx reef_msm_ctx;

typedef struct {
    uint32_t window_bits;   /* Pippenger window c; 0 = choose from n */
    uint32_t bucket_groups; /* G: 0 = one group per window (no precompute); 1 = full precompute
                               (all windows share one bucket set); otherwise windows w and w'
                               share buckets iff w % G == w' % G */
    uint32_t chunk;         /* sorted entries per accumulation thread (L0); 0 = default */
    uint32_t byte_tables;   /* bucket_groups = 1, 1024 < n <= 65536: the sort-free byte tables (256 KiB per point, 14-38 ms to
                               build).  OPT-IN: 0 = the process-wide policy (none, unless REEF_MSM_WIDE=1 is set: then as 3),
                               1 = built inside reef_msm_ctx_create, 2 = never, 3 = built in the background on a
                               low-priority stream of the key's own (the bucket pipeline serves the key until they are
                               ready; calls that run beside the build are slowed by it).  For keys that live long enough
                               to earn them (a prover service).  See reef_msm_ctx_byte_tables. */
    int32_t device;         /* HIP device ordinal; -1 = current device */
    uint32_t reserved[3];
} reef_msm_opts;

reef_status reef_msm_ctx_create(reef_msm_ctx **out, int curve, const reef_affine *bases, size_t n,
                                int bases_loc, const reef_msm_opts *opts /* may be NULL */);
/* 1 when MSMs on this key are served from its byte tables (a mid-size resident key: every signed byte multiple of every point
 * tabulated, an MSM is a plain sum of table entries: no sort, no buckets; 2^15 points 0.19 ms against 0.32 ms), else 0. */
int reef_msm_ctx_byte_tables(reef_msm_ctx *ctx);
/* Replace the key of a ctx in place (same options, workspace kept): what the IPA rounds need, where
 * the generators change every round (CommitmentGens::fold [R], framework.rs:695).  Fails on a ctx
 * whose key is shared with clones. */
reef_status reef_msm_ctx_set_bases(reef_msm_ctx *ctx, const reef_affine *bases, size_t n, int bases_loc);
/* A second handle on the same resident key with its own stream and workspace (for callers that
 * issue MSMs from several threads, e.g. nova's rayon workers inside ipa_pc). */
reef_status reef_msm_ctx_clone(reef_msm_ctx **out, reef_msm_ctx *src);
/* ctx becomes what a clone of src would be -- a handle on src's resident key -- but keeps its own stream and workspace (O(1);
 * waits for the work already enqueued on ctx; same curve and device).  For callers with more keys than threads: one ctx per
 * thread, attached to the key of the moment (what the drop-in symbols do per calling thread). */
reef_status reef_msm_ctx_attach(reef_msm_ctx *ctx, reef_msm_ctx *src);
void reef_msm_ctx_destroy(reef_msm_ctx *ctx);
/* Block until everything enqueued on the ctx's stream has finished. */
reef_status reef_msm_ctx_sync(reef_msm_ctx *ctx);
/* The ctx's hipStream_t (as void*), e.g. to order work of a torch/RCCL stream after it: from here on the ctx stays on this stream of the
 * library's pool.  NULL on failure (no stream could be had; reef_last_error() says why) -- never a stand-in for the default stream. */
void *reef_msm_ctx_stream(reef_msm_ctx *ctx);

/* K1: out = sum_{i<n} scalars[i] * bases[i], n <= key length.
 * Replaces Group::vartime_multiscalar_mul / CE::commit without blind [R] (call sites above).
 * With out_loc == REEF_DEVICE the call only enqueues work (no host sync). */
reef_status reef_msm(reef_msm_ctx *ctx, const reef_fe *scalars, size_t n, int scalars_loc,
                     bool is_mont, reef_jacobian *out, int out_loc);

/* `count` independent MSMs issued at once, MSM i on ctxs[i] with scalars[i][0 .. n[i]): all are enqueued before any is waited
 * for, each on the stream its context takes from the library's pool, so their latency-bound stages overlap on the GPU.  The
 * contexts must be distinct (clone a key that commits twice); they may belong to different curves.  out: `count` commitments in
 * host memory.  For the two commitments of ONE curve inside RecursiveSNARK::prove_step (comm_W of the fresh witness and the
 * cross term comm_T of NIFS::prove [R]: both are absorbed before the folding challenge is drawn, neither needs the other;
 * src/backend/framework.rs:668-675) -- the commitments of the two curves are NOT independent of each other: the secondary
 * step circuit hashes the primary's folded instance, which contains them.  reef_msm_rows(rows = 2) is the other way to issue
 * such a pair (one pass of the pipeline over both; same key, equal lengths). */
reef_status reef_msm_multi(size_t count, reef_msm_ctx *const *ctxs, const reef_fe *const *scalars, const size_t *n, int scalars_loc,
                           bool is_mont, reef_jacobian *out);

/* K2: rows independent MSMs over the same first row_len bases, plus an optional Pedersen blind:
 *   out[r] = sum_j scalars[r*row_len + j] * bases[j]  (+ blinds[r] * h   if blinds != NULL)
 * Replaces HyraxPC::commit(&poly) [R] at src/backend/commitment.rs:187 (L = 2^(l/2) rows of
 * R = 2^(l - l/2) symbols each, src/backend/commitment.rs:173-174) and CE::commit with blind
 * (commitment.rs:350,361,422,430) when rows == 1.  `max_scalar_bits` bounds the canonical
 * scalars (e.g. 8 for ASCII symbols, 3 for DNA; framework.rs:978-1011); 0 = measure on device.
 * blinds/h live where scalars live. */
reef_status reef_msm_rows(reef_msm_ctx *ctx, const reef_fe *scalars, size_t rows, size_t row_len,
                          int scalars_loc, bool is_mont, uint32_t max_scalar_bits,
                          const reef_fe *blinds, const reef_affine *h, reef_jacobian *out,
                          int out_loc);

/* The same commitments from the document's own symbols: one unsigned byte per entry (< 2^symbol_bits,
 * symbol_bits in 1..8), as Reef holds the document before turning it into field elements
 * (src/backend/framework.rs:978-1011 -> NLDocCommitment::new, src/backend/commitment.rs:133-212).
 * Tiny scalars need no bucket method: the generators are taken k at a time (2^(symbol_bits*k) <= 512),
 * every combination is tabulated once per (ctx, row_len, symbol_bits), and a row is the plain sum of
 * row_len/k table entries.  reef_msm_rows takes the same path on its own when max_scalar_bits <= 8 and
 * the batch is large enough to pay for the tables.  blinds/h live where the symbols live; blinds are
 * field elements (Montgomery form if blinds_are_mont). */
reef_status reef_msm_rows_symbols(reef_msm_ctx *ctx, const uint8_t *symbols, size_t rows, size_t row_len, int symbols_loc,
                                  uint32_t symbol_bits, const reef_fe *blinds, const reef_affine *h, bool blinds_are_mont,
                                  reef_jacobian *out, int out_loc);

/* IPA round WITHOUT generator folding.  After k rounds of G'_i = w1*G_i + w2*G_{i+half} the
 * generators are fixed linear combinations of the original ones, so the cross terms of round k
 *     L = <a_lo, G^(k)_hi>,   R = <a_hi, G^(k)_lo>        (a = a_lo || a_hi, n_k = n / 2^k scalars)
 * are two MSMs over the ORIGINAL resident key with scalars a[.] * prod(challenges): this replaces
 * CommitmentGens::fold + the two commits of ipa_pc::InnerProductArgument::prove [R]
 * (CompressedSNARK::prove, src/backend/framework.rs:695; HyraxPC::prove_eval, commitment.rs:371,383)
 * and yields bit-identical L, R.  w1s/w2s: the k challenges so far, canonical integers on the host
 * (same convention as reef_fold); n_k * 2^k must equal the key length.  L and R go to the host. */
reef_status reef_ipa_cross_terms(reef_msm_ctx *ctx, const reef_fe *a, size_t n_k, int a_loc, bool is_mont,
                                 const reef_fe *w1s, const reef_fe *w2s, size_t k, reef_jacobian *out_l,
                                 reef_jacobian *out_r);

/* A commitment over FOLDED generators without folding them: what CE::commit(&gens.fold(..).fold(..), v) [R] returns when
 * `gens` is the resident key of ctx.  With n_k = n / 2^k generators after k folds (challenges w1s[m], w2s[m], canonical
 * integers on the host, same convention as reef_fold),
 *     out = sum_{j < len} v[j] * G^(k)_{off + j},     off + len <= n_k,
 * computed as one MSM over the original key with scalars v[j] * prod(challenges).  `off`/`len` select a slice, so the halves
 * of CommitmentGens::split_at and the last remaining generator (len = 1 after log2(n) folds, v = [1]) are the same call:
 * a CommitmentGens that records its folds instead of performing them (reef_amd/provider.py FoldedGens,
 * host/reef_provider.hpp) serves nova-snark's ipa_pc unchanged -- no 255-bit scalar multiplication per generator, no
 * re-keying.  k = 0 is a commitment over a slice of the key.  The key length must be a multiple of 2^k. */
reef_status reef_msm_folded(reef_msm_ctx *ctx, const reef_fe *v, size_t len, size_t off, int v_loc, bool is_mont,
                            const reef_fe *w1s, const reef_fe *w2s, size_t k, reef_jacobian *out, int out_loc);

/* ---------------------------------------------------------------------------------------------
 * (3) Stateless helpers around the MSMs.
 * ------------------------------------------------------------------------------------------- */
/* K3: out[i] = w1*gens[i] + w2*gens[half+i], i < half, affine out.
 * Replaces CommitmentGens::fold [R] as used by ipa_pc::InnerProductArgument::prove inside
 * CompressedSNARK::prove (src/backend/framework.rs:695).  w1, w2: canonical (non-Montgomery)
 * 32-byte little-endian scalars on the host. */
reef_status reef_fold(int curve, const reef_affine *gens, size_t half, int loc, const reef_fe *w1,
                      const reef_fe *w2, reef_affine *out);

/* K4: batch Jacobian -> affine and/or 32-byte compressed encoding (LE canonical x, y parity in
 * bit 255, identity = zeros).  Replaces Commitment::compress / to_affine [R]
 * (src/backend/commitment.rs:195,351,365,425,427,431).  Either output may be NULL. */
reef_status reef_normalize(int curve, const reef_jacobian *in, size_t n, int loc, reef_affine *out_affine,
                           uint8_t *out_compressed);

/* Sum of n Jacobian points (multi-GPU: combine the per-rank partial MSMs after an all-gather). */
reef_status reef_sum_points(int curve, const reef_jacobian *in, size_t n, int loc, reef_jacobian *out);

/* Deterministic synthetic inputs, generated on the device (bench / tests; no file I/O):
 * bases B_i = (k0 + i*d)*G with G = (-1, 2); scalars from a SplitMix64 stream
 * (kind 0 uniform, 1 witness-like 70/20/10 mix, 2 uniform below small_bound). */
reef_status reef_gen_bases(int curve, uint64_t k0, uint64_t d, size_t n, reef_affine *out, int loc);
reef_status reef_gen_scalars(int curve, uint64_t seed, int kind, uint64_t small_bound, size_t n,
                             bool to_mont, reef_fe *out, int loc);

/* ---------------------------------------------------------------------------------------------
 * (3b) Row N2 ("next" in SURVEY.md 8f): the host sum-check of nlookup witness generation as
 * vector kernels over the scalar field of `curve` (Reef: REEF_PALLAS, i.e. Fq, the CirC modulus of
 * src/backend/r1cs_helper.rs:37-38).  All values cross the ABI as canonical integers, 32 bytes
 * little-endian -- what the reference's rug::Integer tables hold -- NOT in Montgomery form.
 * Replaces, per folding step (src/backend/r1cs.rs:2318-2385):
 *   gen_eq_table              r1cs_helper.rs:508-544  -> reef_sc_gen_eq_table
 *   linear_mle_product        r1cs_helper.rs:441-506  -> reef_sc_round_coeffs (sums, :455-476), then
 *                                                        reef_sc_fold (both tables, :491-503) with the
 *                                                        Poseidon challenge the host derived (:478-489)
 *   prover_mle_partial_eval(table, sc_rs)  :551-634   -> after the last round the folded table holds
 *                                                        the value in entry 0: reef_sc_read(ctx, 0, 1, ..)
 * The results are the reference's whatever happens inside; what happens inside when the calls come in the reference's
 * order (reset, gen_eq_table, rounds with halving pow): the EQ table of gen_eq_table is a rank-one table (two factor
 * tables) plus the lookup points and is kept as that while the rounds fold its high index bits -- nothing of 2^ell
 * entries is written for it, a round streams T only; and the first round of a step reads T through the row structure
 * reef_sc_set_table found (constant rows, rows of small entries such as document symbols).  A caller-given EQ
 * (which = 1), reef_sc_read(ctx, 1, ..) between rounds, or a round out of order: dense tables from there on.
 * ------------------------------------------------------------------------------------------- */
typedef struct reef_sc_ctx reef_sc_ctx;
/* Two resident tables (T = lookup table or document, EQ) of table_len = 2^ell entries each. */
reef_status reef_sc_create(reef_sc_ctx **out, int curve, size_t table_len);
void reef_sc_destroy(reef_sc_ctx *ctx);
/* which: 0 = T, 1 = EQ.  n <= table_len values; the rest is zero-padded (r1cs.rs:2323-2330). */
reef_status reef_sc_set_table(reef_sc_ctx *ctx, int which, const reef_fe *values, size_t n, int loc);
/* EQ[i] = sum_{k: qs[k] = i} rs[k] + rs[nq] * prod_j (bit_j(i) ? last_q[j] : 1 - last_q[j]);
 * rs has nq + 1 entries, last_q has ell entries (host arrays). */
reef_status reef_sc_gen_eq_table(reef_sc_ctx *ctx, const reef_fe *rs, const uint32_t *qs, size_t nq,
                                 const reef_fe *last_q, size_t ell);
/* Round with pow = 2^(ell - i): out = { xsq, x, con } (host). */
reef_status reef_sc_round_coeffs(reef_sc_ctx *ctx, size_t pow, reef_fe out[3]);
/* X[b] <- X[b]*(1 - r) + X[b + pow]*r for b < pow, both tables (asynchronous).  Only entries [0, pow) are
 * written: after a fold the tables hold `pow` live entries each, and reads or rounds that reach beyond them
 * are rejected with REEF_ERR_ARG until the table is set / reset (T) or generated (EQ) again. */
reef_status reef_sc_fold(reef_sc_ctx *ctx, size_t pow, const reef_fe *r);
/* reef_sc_fold(pow, r) and reef_sc_round_coeffs(pow / 2) in ONE pass over the tables (pow >= 2):
 * the fold of round i feeds the sums of round i+1 from registers. */
reef_status reef_sc_fold_and_next_coeffs(reef_sc_ctx *ctx, size_t pow, const reef_fe *r, reef_fe out[3]);
/* Read back the first `count` entries of a table as canonical integers (host). */
reef_status reef_sc_read(reef_sc_ctx *ctx, int which, size_t count, reef_fe *out);
/* T <- the values last given to reef_sc_set_table(ctx, 0, ..): every folding step starts from the
 * unfolded table (the reference clones it per step, r1cs.rs:2320); device-to-device, asynchronous. */
reef_status reef_sc_reset_table(reef_sc_ctx *ctx);
reef_status reef_sc_sync(reef_sc_ctx *ctx);

/* ---------------------------------------------------------------------------------------------
 * (3c) Row N3: the O(N) field passes over the committed document at proof end, over the SCALAR
 *      field of `curve`.
 *
 * Replaces, in NLDocCommitment::proof_dot_prod_prover (src/backend/commitment.rs:287-405):
 *   doc_poly.evaluate(&running_q)                               commitment.rs:357
 *   the bound rows LZ = L^T Z inside hyrax_gen.prove_eval(..)   commitment.rs:371-379, :383-391
 *   (the dot-product IPA over LZ that follows runs on reef_msm / reef_ipa_cross_terms)
 * and verifier_mle_eval(table, q') of prove_consistency (commitment.rs:236; r1cs_helper.rs:637-641).
 *
 * z: the table, n entries (n <= 2^num_vars; the rest is zero padding), row-major as the
 * 2^left_vars x 2^(num_vars-left_vars) matrix Hyrax commits to (compute_factored_lens,
 * commitment.rs:173-174).  elem_bytes = 32: field elements in the same form as `point`
 * (is_mont: pasta ABI Montgomery form, else canonical integers); elem_bytes = 1, 2 or 4: unsigned
 * little-endian document symbols (framework.rs:978-1011).  point[0] pairs with the most
 * significant index bit (r1cs_helper.rs:577-592).
 *
 *   lz_out[j]  = sum_i eq(point[..left_vars], i) * z[i * 2^(num_vars-left_vars) + j]
 *   *eval_out  = sum_j lz[j] * eq(point[left_vars..], j)     (= the multilinear extension at point)
 *
 * Either output may be NULL.  Outputs are in the form `is_mont` names; eval_out is host memory.
 * Blind combination sum_i L_i * blind_i: the same call with the blinds as a one-column table
 * (num_vars = left_vars). */
reef_status reef_mle_bound_rows(int curve, const void *z, size_t n, int elem_bytes, int z_loc, bool is_mont,
                                const reef_fe *point, size_t num_vars, size_t left_vars, reef_fe *lz_out, int out_loc,
                                reef_fe *eval_out);

/* ---------------------------------------------------------------------------------------------
 * (3d) Row N4: the Poseidon Merkle commitment of the document (`--merkle`).
 *
 * Replaces MerkleCommitment::new(&doc, &pc) (src/backend/merkle_tree.rs:25-80, new_parent :82-114; built at
 * src/backend/commitment.rs:94-100) over the SCALAR field of `curve` (Reef: REEF_PALLAS, G1::Scalar):
 *   level 0, node i = H4(2i, doc[2i], 2i+1, doc[2i+1])      an odd last symbol hashes (2i, doc[2i], 0, 0)
 *   level h, node i = H2(below[2i], below[2i+1])             an odd last node hashes (below[2i], 0)
 *   H_k(x_1..x_k)   = P([tag_k, x_1, .., x_k, 0, ..])[1]     P = the Poseidon permutation of width 5 (x^5 S-box,
 *                     full_rounds/2 full rounds, partial_rounds partial rounds, full_rounds/2 full rounds; a round adds
 *                     the round constants, applies the S-box and multiplies by the MDS matrix)
 * which is what neptune's Sponge<F, U4> in Mode::Simplex computes for IOPattern [Absorb(k), Squeeze(1)] [R].  The
 * constants are the CALLER'S: round_constants, mds and the two tags are neptune's PoseidonConstants fields and the
 * domain tags its sponge API derives from the two IO patterns -- that crate is not in the reference tree, so nothing
 * here restates it.  All field elements (constants in, tree out) are canonical integers, or in the pasta Montgomery
 * form when is_mont.  `doc`: one 32-bit value per document symbol (the indices of framework.rs:978-1011).
 * tree_out (may be NULL): all levels, level 0 first, reef_merkle_nodes(n) elements; root_out (host, may be NULL). */
typedef struct {
    uint32_t width;              /* state width t = arity + 1; only 5 (Reef's U4) is built */
    uint32_t full_rounds;        /* R_F, even */
    uint32_t partial_rounds;     /* R_P */
    uint32_t reserved;
    const reef_fe *round_constants; /* host: width * (R_F + R_P) elements, round-major */
    const reef_fe *mds;          /* host: width * width; new[j] = sum_i state[i] * mds[i*width + j] */
    reef_fe tag_leaf;            /* state[0] of the 4-input leaf hash */
    reef_fe tag_node;            /* state[0] of the 2-input node hash */
} reef_poseidon_params;
uint64_t reef_merkle_nodes(uint64_t n);
reef_status reef_merkle_commit(int curve, const reef_poseidon_params *params, const uint32_t *doc, size_t n, int doc_loc,
                               bool is_mont, reef_fe *tree_out, int tree_loc, reef_fe *root_out);
/* The same tree built by several GPUs of ONE process (BASELINE configs[4]: the 64 MiB --merkle document on 8 GPUs; SURVEY.md 8e.1:
 * independent units).  The bottom level is cut into blocks of 2^L nodes -- the smallest L that leaves at most ndev blocks -- and block b is
 * hashed by devices[b] on a host thread of its own: a block is a subtree of the whole tree (its leaf hashes take the symbols' positions in
 * the whole document; a ragged last block keeps hashing (node, 0) up to level L as the whole tree does), so the devices exchange nothing
 * but their block's root, 32 bytes each, through the host; the levels above L are hashed from those roots on devices[0].  doc, tree_out
 * (may be NULL) and root_out (may be NULL, not both) are HOST memory: every device reads its slice of the document and writes its slices of
 * the levels over its own PCIe link.  Node for node the tree of reef_merkle_commit.  *blocks_out (may be NULL): how many blocks, hence
 * devices, the call used (a power-of-two cut: 3 devices and 2^26 bottom nodes give 2 blocks).  devices[] may repeat an ordinal. */
reef_status reef_merkle_commit_devices(int curve, const reef_poseidon_params *params, const uint32_t *doc, size_t n, bool is_mont,
                                       const int *devices, size_t ndev, reef_fe *tree_out, reef_fe *root_out, uint32_t *blocks_out);

/* ---------------------------------------------------------------------------------------------
 * (3e) Row N1: derivation of a commitment key from a label.
 *
 * Replaces nova-snark's CommitmentGens::new(label, n) -> from_label [R] (called at PublicParams::setup,
 * src/backend/framework.rs:297-303, and at SpartanSNARK::setup / HyraxPC::setup, src/backend/commitment.rs:146-149,
 * 176-180; once per --prove and once per --verify):
 *   stream  = SHAKE256(label), squeezed into n strings of 32 bytes                    (host: the XOF is sequential)
 *   out[i]  = hash_to_curve(stream[32 i .. 32 i + 32))                                (GPU: one generator per thread)
 * hash_to_curve is RFC 9380's: u0, u1 = hash_to_field(msg, 2) with expand_message_xmd over BLAKE2b-512 and 64-byte
 * strings per element; Q_j = iso_map(map_to_curve_simple_swu(u_j)) through the curve E': y^2 = x^3 + a x + b that is
 * 3-isogenous to y^2 = x^3 + 5; out = Q_0 + Q_1 (the Pasta curves have cofactor 1).  Everything pasta_curves fixes
 * and this repository cannot confirm from the reference tree [R] is DATA of the call: a, b, Z of the SWU map, the 13
 * coefficients of the isogeny (x_num[4], x_den[2], y_num[4], y_den[3]: highest degree first, the monic leading terms
 * of the denominators left out), the domain separation string and the byte order hash_to_field reads its 64-byte
 * strings in.  Field elements of the parameters are canonical integers of the curve's BASE field, or pasta
 * Montgomery form when is_mont.  out: n affine points in the ABI form (identity = (0, 0)), host or device. */
typedef struct {
    reef_fe a, b, z;             /* E' and the SWU constant Z (a non-square with g(b/(Z a)) square, RFC 9380 6.6.2) */
    reef_fe iso[13];             /* x_num[4], x_den[2], y_num[4], y_den[3] */
    const uint8_t *dst;          /* host: domain separation string, at most 255 bytes */
    uint32_t dst_len;
    uint32_t little_endian;      /* 0: OS2IP big-endian strings (RFC 9380 5.2); 1: little-endian */
} reef_keygen_params;
reef_status reef_derive_generators(int curve, const uint8_t *label, size_t label_len, size_t n, const reef_keygen_params *params,
                                   bool is_mont, reef_affine *out, int out_loc);
/* SHAKE256(in) -> out_len bytes: the host half of the derivation, exported for tests and for callers that want the stream. */
void reef_shake256(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);

/* ---------------------------------------------------------------------------------------------
 * (4) Runtime plumbing.
 * ------------------------------------------------------------------------------------------- */
int reef_device_count(void);
reef_status reef_set_device(int ordinal);
reef_status reef_get_device(int *ordinal);              /* the calling thread's current device */
reef_status reef_device_sync(void);
void *reef_device_alloc(size_t bytes);                 /* hipMalloc; NULL on failure */
void reef_device_free(void *p);
reef_status reef_memcpy(void *dst, const void *src, size_t bytes, int dst_loc, int src_loc);
const char *reef_last_error(void);                     /* thread-local message of the last failure */
const char *reef_version(void);
uint32_t reef_abi_version(void);                       /* REEF_ABI_VERSION of the library that was loaded */
/* Process-wide runtime settings.  The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless set) and
 * streams that share a queue run in turn; concurrent callers of this library (the three arguments of the final SNARK, rayon workers:
 * src/backend/framework.rs:695-721) want 8.  The variable is read at the process's FIRST HIP call, so this must run before it.
 * OPT-IN (ABI 5): the library touches the environment only when asked -- hw_queues > 0 here, or REEF_MSM_HW_QUEUES=<n> exported when
 * the library is loaded; hw_queues < 0 withdraws a value the library set, 0 changes nothing.  A value the user exported as
 * GPU_MAX_HW_QUEUES is never overwritten.  info (may be NULL) reports what the environment holds and who put it there.
 *
 * warm (ABI 6): a process's first calls into HIP pay for the runtime, not for the work -- measured on a bare caller of mult_pippenger_pallas
 * (profiles/r06_seam_hip_first_use.txt): runtime initialisation 51 ms, the first stream 21 ms, the first launch of a code object 7 ms (this library
 * carries one per curve), the first copy in each direction 8 ms: ~100 ms before the first commitment of a proof whose commitments take 20-30 ms in
 * all.  Reef spends far longer than that on the host before its first MSM (regex -> SAFA, the step circuit: src/backend/framework.rs:81-166), so the
 * warm-up can hide behind it: REEF_WARM_BACKGROUND starts a thread of the library that initialises the runtime on the calling thread's CURRENT
 * device (device 0 for a thread that never chose one), creates the pool's first stream, launches one kernel of each curve's code object and moves a
 * few bytes host -> device -> host; the call returns at once, and a library call that arrives before the thread has finished simply runs beside it
 * (HIP's own locks order them).  REEF_WARM_NOW does the same before it returns.  REEF_MSM_WARM=1 in the environment asks for the background form
 * when the library is LOADED -- the zero-patch route, which has no init call.  info->warm reports 0 (never asked), 1 (running), 2 (done), 3 (failed:
 * no usable device -- the first real call will say why). */
enum { REEF_WARM_NONE = 0, REEF_WARM_NOW = 1, REEF_WARM_BACKGROUND = 2 };
typedef struct {
    int32_t hw_queues;
    uint32_t warm;                     /* REEF_WARM_* */
    uint32_t reserved[6];
} reef_runtime_opts;
typedef struct {
    int32_t hw_queues_env;             /* GPU_MAX_HW_QUEUES as the environment holds it now (0: unset) */
    int32_t hw_queues_set_by_library;  /* the value this library put there (0: it did not) */
    uint32_t abi_version;
    uint32_t warm;                     /* 0 never asked, 1 running, 2 done, 3 failed */
} reef_runtime_info;
reef_status reef_runtime_init(const reef_runtime_opts *opts /* may be NULL */, reef_runtime_info *info /* may be NULL */);

/* One MSM split by Pippenger window across `world` GPUs (north_star): every GPU holds
 * the whole key and receives all scalars, but accumulates only the windows w = rank (mod world), so
 * reef_msm / reef_msm_rows on this ctx return a PARTIAL sum; the N partial sums (96 B each) are
 * exchanged (RCCL all-gather) and added (reef_msm_ctx_sum_points).  world = 1 restores whole MSMs.
 * A Pedersen blind term is added by rank 0 only, and row commitments that go through the symbol tables
 * (no windows to split) are computed whole by rank 0, the other ranks returning the identity.
 * Clones made afterwards inherit the setting. */
reef_status reef_msm_ctx_set_window_split(reef_msm_ctx *ctx, uint32_t rank, uint32_t world);
/* Per-MSM HIP-event timing is opt-in (each event record costs ~6 us of stream time). */
reef_status reef_msm_ctx_enable_timing(reef_msm_ctx *ctx, int on);
/* Timing of the last reef_msm / reef_msm_rows on this ctx, measured with HIP events on the ctx's
 * stream (valid after a sync): total and the accumulation kernel alone, in milliseconds. */
reef_status reef_msm_ctx_last_timing(reef_msm_ctx *ctx, float *total_ms, float *accumulate_ms);
/* The same, summed over every MSM issued on this ctx since the last reset: number of calls, total
 * milliseconds and milliseconds inside the bucket-accumulation kernel (HIP events recorded on the
 * ctx's stream around each launch; the call waits for MSMs still in flight). */
reef_status reef_msm_ctx_timing_stats(reef_msm_ctx *ctx, int reset, uint64_t *calls, double *total_ms,
                                      double *accumulate_ms);
/* Sum of n Jacobian points, enqueued on the ctx's stream (device buffers only): combines the
 * per-rank partial MSMs after an RCCL all-gather ordered on the same stream. */
reef_status reef_msm_ctx_sum_points(reef_msm_ctx *ctx, const reef_jacobian *in, size_t n, reef_jacobian *out);
/* Plan actually used by the ctx: window bits, windows, bucket groups, tables. */
reef_status reef_msm_ctx_plan(reef_msm_ctx *ctx, uint32_t *c, uint32_t *windows, uint32_t *groups,
                              uint32_t *tables);

/* Plan the engine would choose for an n-point key with the given options (pure host logic; works
 * without a GPU): window bits c, windows W = ceil(256/c), bucket groups G, tables T = ceil(W/G). */
reef_status reef_msm_plan_for(size_t n, uint32_t window_bits, uint32_t bucket_groups, uint32_t *c,
                              uint32_t *windows, uint32_t *groups, uint32_t *tables);

/* Device self-tests used by the parity suite (element-wise kernels over n inputs, HOST buffers).
 * field: 0 Fp, 1 Fq.  op: 0 mul 1 add 2 sub 3 inv 4 to_mont 5 from_mont 6 neg 7 sqr. */
reef_status reef_test_field_op(int field, int op, const reef_fe *a, const reef_fe *b, reef_fe *out, size_t n);
/* op: 0 mixed add P+Q, 1 general add, 2 double P, 3 k*P (k canonical in kbuf); the four-wave forms of the tail
 * kernels: 4 general add, 5 double, 6 the chain 4*(P + Q) + P, 7-10 latency probes, 11-19 every back-to-back order
 * of additions and doublings, 20 a doubling followed by an addition that falls into its own doubling (4P), 21 an addition
 * of Z = 1 operands, 22 a step machine with one addition site and one doubling site (7*(P + Q)). */
reef_status reef_test_ec_op(int curve, int op, const reef_affine *p, const reef_affine *q, const reef_fe *k,
                            reef_jacobian *out, size_t n);
/* Field-multiplication throughput probe: returns Montgomery products per second. */
reef_status reef_bench_fmul(int field, uint32_t iters, double *products_per_s);

/* ---------------------------------------------------------------------------------------------
 * (5) Device groups: the multi-GPU split behind the C ABI, in ONE process.
 *
 * Reef's prover is one Rust process (src/backend/main.rs:82; src/backend/framework.rs:81-166: two OS threads, no process
 * boundary), so what BASELINE's north_star asks for -- "large MSMs are split by Pippenger window across the 8 GPUs of one node
 * with a reduce of the partial sums over xGMI" -- has to be reachable from one address space: a group owns one resident-key
 * context per member device, issues a call on every member before it waits for any (one host thread per member: eight PCIe links
 * carry the scalars at once), brings the 96-byte partial sums together on member 0's device and adds them there.
 *
 *   REEF_SPLIT_WINDOWS  every member holds the WHOLE key (members that share a device share one copy) and receives all the
 *                       scalars; member i accumulates the Pippenger windows w = i (mod ndev) (reef_msm_ctx_set_window_split):
 *                       the split north_star names.  The rows of reef_msm_group_rows are dealt out whole on such a group.
 *   REEF_SPLIT_POINTS   member i holds points [i*n/ndev, (i+1)*n/ndev) of the key and receives that slice of every scalar
 *                       vector: 1/ndev of the key memory and of the upload per device (SURVEY.md 8e.2 "by points").
 *
 * Exchange (reef_msm_group_info.exchange): REEF_EXCHANGE_PEER -- a member on another device than member 0 sends its partial sum
 * with hipMemcpyPeerAsync on its own stream (xGMI between the GPUs of one node; peer access is enabled where the devices allow
 * it), a member on member 0's device writes it in place; member 0's stream waits for the members' events and runs the sum
 * kernel (RCCL has no elliptic-curve reduction and the payload is 96 bytes per member: latency, not bandwidth).
 * REEF_EXCHANGE_HOST -- the labelled fallback: every member's last kernel stores into its slot of host-mapped pinned memory,
 * the host waits for all members and the slots are summed on member 0's device.  gopts->exchange = 0 takes PEER.
 * REEF_EXCHANGE_RCCL -- the exchange north_star words ("an RCCL reduce of the partial sums over xGMI"), from one process: a
 * single-process communicator over the group's DISTINCT devices (ncclCommInitAll at reef_msm_group_create; RCCL is loaded at
 * run time -- librccl.so.1, or the file REEF_RCCL_LIB names -- so the library carries no link-time dependency on it, and the
 * creation FAILS, with the loader's message, where it cannot be loaded: no silent fallback).  Per call: one ncclGroup of
 * ncclSend (member i's 96 bytes, on its device's stream) / ncclRecv (member 0's device, into slot i) pairs, then the same sum
 * kernel.  RCCL offers no reduction over curve points, so the "reduce" is a gather plus k_sum_points whichever exchange is
 * chosen.  Every member but member 0 goes through a send/receive pair, also a member that shares member 0's device (a rank
 * sending to itself): a one-GPU box runs the calls an eight-GPU node runs, with other peers.
 *
 * STATUS: everything in this section has run on ONE physical device only -- groups whose members repeat ordinal 0.  The
 * branches that need two devices (hipMemcpyPeerAsync between different devices, peer-access enablement, events waited for across devices,
 * RCCL send/recv between different ranks) are HIP's and RCCL's documented calls and have never executed here: treat the section as
 * EXPERIMENTAL until it has met a multi-GPU node; reef_msm_group_enable_timing exists so that the first such run explains itself.
 *
 * devices[] may REPEAT an ordinal: a group of 2 / 3 / 8 members on device 0 runs every code path on a one-GPU box (that is how
 * tests/test_gpu_group.py covers it); members that share a device share the resident key (clones).  Results are identical to
 * reef_msm / reef_msm_rows on one context, whatever the split.  A group serialises the calls made on it.
 * Replaces, as (2) does on one device: nova-snark's CommitmentGens<G> + CE::commit [R] at src/backend/framework.rs:668-721 and
 * HyraxPC::commit at src/backend/commitment.rs:187.
 * ------------------------------------------------------------------------------------------- */
typedef struct reef_msm_group reef_msm_group;
enum { REEF_SPLIT_WINDOWS = 0, REEF_SPLIT_POINTS = 1 };
enum { REEF_EXCHANGE_DEFAULT = 0, REEF_EXCHANGE_PEER = 1, REEF_EXCHANGE_HOST = 2, REEF_EXCHANGE_RCCL = 3 };
/* How HOST scalars reach the members of a REEF_SPLIT_WINDOWS group (every member needs ALL of them: 8 x 32 MiB at 2^20 points).
 * REEF_SCALARS_EACH (default): every member uploads them from the caller's memory over its own PCIe link, all links at once.
 * REEF_SCALARS_FANOUT: ONE upload to devices[0], then every other member fetches them from there (hipMemcpyPeerAsync on its own stream, after
 * the upload's event: xGMI, 7 links x ~153 GB/s per GPU against one PCIe link per GPU shared with nothing) -- every member but member 0 goes
 * through the copy, also one that shares devices[0], so a one-GPU box runs the calls a node runs.  Which of the two wins depends on the host's
 * PCIe topology and on how much of the caller's memory is pageable; bench.py --single-process reports both.  Ignored for device scalars
 * (already a fan-out from devices[0]) and for points groups (every member uploads its own slice only). */
enum { REEF_SCALARS_EACH = 0, REEF_SCALARS_FANOUT = 1 };
typedef struct {
    uint32_t split;        /* REEF_SPLIT_* */
    uint32_t exchange;     /* REEF_EXCHANGE_* */
    uint32_t scalars;      /* REEF_SCALARS_* */
    uint32_t reserved[5];
} reef_msm_group_opts;
typedef struct {
    uint32_t members;          /* ndev */
    uint32_t distinct_devices; /* how many different ordinals devices[] named */
    uint32_t split, exchange;  /* as resolved */
    uint32_t peer_members;     /* members on another device than member 0 whose device has direct peer access to it (xGMI) */
    uint32_t reserved[3];
    uint64_t key_points[16];   /* points of the key resident on each of the first 16 members */
} reef_msm_group_info;
/* bases: the whole key, n points, on the host or on devices[0] (bases_loc).  key_opts as for reef_msm_ctx_create (its `device`
 * field is ignored: devices[] decides); 1 <= ndev <= 64. */
reef_status reef_msm_group_create(reef_msm_group **out, int curve, const reef_affine *bases, size_t n, int bases_loc,
                                  const reef_msm_opts *key_opts /* may be NULL */, const int *devices, size_t ndev,
                                  const reef_msm_group_opts *gopts /* may be NULL: windows, peer */);
void reef_msm_group_destroy(reef_msm_group *grp);
reef_status reef_msm_group_info_get(reef_msm_group *grp, reef_msm_group_info *info);
/* Where the time of a group call goes (diagnostics; round 6: the first run on several physical devices cannot be rehearsed on a one-GPU box, so
 * one call must be able to say which term is off).  With timing enabled a split call (reef_msm_group_msm, reef_msm_group_rows with rows == 1)
 * waits for every member separately before the partial sums are added -- a few microseconds of lost overlap -- and records, for the LAST call
 * (milliseconds, host clock unless said otherwise):
 *   total_ms         call entry -> result on the host
 *   distribute_ms    call entry -> every member's share enqueued (the slowest member's issue call: staging of the caller's pageable scalars,
 *                    peer fetches of device scalars, kernel launches)
 *   members_done_ms  call entry -> every member's last kernel and the send of its partial sum finished
 *   combine_ms       from there -> the sum of the partial sums on the host (k_sum_points on devices[0], one wait)
 *   member_issue_ms[i]   member i's issue call alone
 *   member_stream_ms[i]  member i's time on its stream, first enqueue to the send of its partial sum (HIP events): its share of the scalars'
 *                        way to the device, its MSM, its send
 * members beyond the 16th are not itemised. */
typedef struct {
    uint32_t members, reserved;
    double total_ms, distribute_ms, members_done_ms, combine_ms;
    double member_issue_ms[16], member_stream_ms[16];
} reef_msm_group_timing;
reef_status reef_msm_group_enable_timing(reef_msm_group *grp, int on);
reef_status reef_msm_group_last_timing(reef_msm_group *grp, reef_msm_group_timing *out);
/* K1 over the group: out (HOST) = sum_{i<n} scalars[i] * key[i], n <= key length.  scalars: host memory, or device memory of
 * devices[0] (the other devices fetch their share with peer copies). */
reef_status reef_msm_group_msm(reef_msm_group *grp, const reef_fe *scalars, size_t n, int scalars_loc, bool is_mont, reef_jacobian *out);
/* K2 over the group (REEF_SPLIT_WINDOWS groups: every member needs the first row_len points): reef_msm_rows with the rows dealt
 * out in contiguous blocks, member i computing rows [i*rows/ndev, (i+1)*rows/ndev) whole and writing them straight into out
 * (HOST): independent units, no exchange.  rows == 1 -- CE::commit with a blind -- is split by window like
 * reef_msm_group_msm, the blind term added by member 0 only.  Arguments as reef_msm_rows; scalars / blinds / h on the host or on
 * devices[0]. */
reef_status reef_msm_group_rows(reef_msm_group *grp, const reef_fe *scalars, size_t rows, size_t row_len, int scalars_loc, bool is_mont,
                                uint32_t max_scalar_bits, const reef_fe *blinds, const reef_affine *h, reef_jacobian *out);
/* The same from one-byte document symbols (reef_msm_rows_symbols). */
reef_status reef_msm_group_rows_symbols(reef_msm_group *grp, const uint8_t *symbols, size_t rows, size_t row_len, int symbols_loc,
                                        uint32_t symbol_bits, const reef_fe *blinds, const reef_affine *h, bool blinds_are_mont,
                                        reef_jacobian *out);

#ifdef __cplusplus
}
#endif
#endif /* REEF_MSM_H */
