// C++ host-side mirror of the nova-snark provider interface as eniac/Reef calls it, over the C ABI
// of include/reef_msm.h (header only).  Reef is Rust; this image has no Rust toolchain, so the host
// side above the ABI is written in C++ with the reference's names and argument meaning:
//
//   CommitmentGens<CURVE>::commit(v, blind)       CE::commit(&gens, &v, &blind)          src/backend/commitment.rs:349-351,360-361,422,430
//   CommitmentGens<CURVE>::fold(w1, w2)           CommitmentGens::fold (ipa_pc)          reached from src/backend/framework.rs:695
//   CommitmentGens<CURVE>::ipa_cross_terms(..)    the two cross-term commitments of an IPA round, without folding the generators
//   CommitmentGens<CURVE>::commit_folded(..)      any commitment over folded generators (a slice of them), folds recorded not performed
//   HyraxPC<CURVE>::commit / commit_symbols       HyraxPC::commit(&poly)                 src/backend/commitment.rs:187
//   HyraxPC<CURVE>::bind_rows                     first step of HyraxPC::prove_eval      src/backend/commitment.rs:371-391 (and :357)
//   CommitmentGensOnDevices<CURVE>                the same key on several GPUs of this process (device groups, reef_msm.h section 5):
//       ::commit / ::commit_rows / ::commit_symbols   CE::commit split by Pippenger window over the devices, HyraxPC::commit rows dealt out whole
//   SumCheck                                      gen_eq_table / linear_mle_product      src/backend/r1cs_helper.rs:441-544, driven as in r1cs.rs:2318-2385
//   compress()                                    Commitment::compress                   src/backend/commitment.rs:195,351,365
//   CommitmentGens<CURVE>(label, n, params [, h]) CommitmentGens::new(label, n) / new_with_blinding_gen      src/backend/framework.rs:297-303, commitment.rs:146-149,176-180
//   MerkleCommitment<CURVE>(doc, pc)              MerkleCommitment::new(&doc, &pc), path_wits, make_wits     src/backend/merkle_tree.rs:25-190
//
// Error behaviour: Reef treats every failure as a panic (framework.rs:683,702); here every failed
// call throws reef_provider::Error carrying reef_last_error().
#pragma once
#include <array>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "reef_msm.h"

namespace reef_provider {

struct Error : std::runtime_error {
    reef_status status;
    Error(reef_status s, const char *what_failed) : std::runtime_error(std::string(what_failed) + ": " + reef_last_error()), status(s) {}
};
inline void check(reef_status s, const char *what) {
    if (s != REEF_OK) throw Error(s, what);
}

using Compressed = std::array<uint8_t, 32>;

template <int CURVE> inline Compressed compress(const reef_jacobian &p) {
    Compressed c;
    check(reef_normalize(CURVE, &p, 1, REEF_HOST, nullptr, c.data()), "reef_normalize");
    return c;
}

// nova-snark CommitmentGens<G>: a vector of generators (+ blinding generator h) resident on the GPU,
// pre-shifted once (commitment keys are fixed for the life of PublicParams, framework.rs:45,297-303).
template <int CURVE> class CommitmentGens {
  public:
    CommitmentGens(const reef_affine *gens, size_t n, int gens_loc = REEF_HOST, const reef_affine *h = nullptr, bool preshift = true)
        : n_(n), has_h_(h != nullptr) {
        reef_msm_opts o = {};
        o.bucket_groups = preshift ? 1 : 0;
        o.device = -1;
        check(reef_msm_ctx_create(&ctx_, CURVE, gens, n, gens_loc, &o), "reef_msm_ctx_create");
        if (h) h_ = *h;
    }
    // CommitmentGens::new(label, n) [R] / the fork's new_with_blinding_gen(label, n, &h): the n generators are derived from the label on the GPU
    // (row N1: reef_derive_generators) straight into the resident key -- they never visit the host.  pasta_curves' hash-to-curve constants are
    // the caller's (`kp`: canonical integers of the base field unless kp_is_mont).
    CommitmentGens(const std::string &label, size_t n, const reef_keygen_params &kp, const reef_affine *h = nullptr, bool kp_is_mont = false, bool preshift = true)
        : n_(n), has_h_(h != nullptr) {
        reef_affine *dev = (reef_affine *)reef_device_alloc(n * sizeof(reef_affine));
        if (!dev) throw Error(REEF_ERR_OOM, "reef_device_alloc");
        reef_status st = reef_derive_generators(CURVE, (const uint8_t *)label.data(), label.size(), n, &kp, kp_is_mont, dev, REEF_DEVICE);
        if (st == REEF_OK) {
            reef_msm_opts o = {};
            o.bucket_groups = preshift ? 1 : 0;
            o.device = -1;
            st = reef_msm_ctx_create(&ctx_, CURVE, dev, n, REEF_DEVICE, &o);
        }
        reef_device_free(dev);
        check(st, "CommitmentGens::new(label, n)");
        if (h) h_ = *h;
    }
    // the same generators on the host (what from_label returns to a caller that keeps them itself)
    static std::vector<reef_affine> from_label(const std::string &label, size_t n, const reef_keygen_params &kp, bool kp_is_mont = false) {
        std::vector<reef_affine> out(n);
        check(reef_derive_generators(CURVE, (const uint8_t *)label.data(), label.size(), n, &kp, kp_is_mont, out.data(), REEF_HOST), "reef_derive_generators");
        return out;
    }
    ~CommitmentGens() { reef_msm_ctx_destroy(ctx_); }
    CommitmentGens(const CommitmentGens &) = delete;
    CommitmentGens &operator=(const CommitmentGens &) = delete;

    size_t len() const { return n_; }
    reef_msm_ctx *handle() const { return ctx_; }
    const reef_affine *h() const { return has_h_ ? &h_ : nullptr; }

    // CE::commit: sum v_i G_i (+ blind * h).  Scalars in the pasta ABI form (Montgomery), host or device.
    reef_jacobian commit(const reef_fe *v, size_t n, const reef_fe *blind = nullptr, int v_loc = REEF_HOST) const {
        reef_jacobian out;
        if (blind) {
            if (!has_h_) throw std::logic_error("commit with a blind needs the blinding generator");
            if (v_loc != REEF_HOST) throw std::logic_error("blind and h live where the scalars live: host");
            check(reef_msm_rows(ctx_, v, 1, n, REEF_HOST, true, 0, blind, &h_, &out, REEF_HOST), "reef_msm_rows");
        } else {
            check(reef_msm(ctx_, v, n, v_loc, true, &out, REEF_HOST), "reef_msm");
        }
        return out;
    }

    // CommitmentGens::fold(w1, w2): G'_i = w1 * G_i + w2 * G_{i + n/2}; w canonical little-endian.
    static std::vector<reef_affine> fold(const reef_affine *gens, size_t n, const reef_fe &w1, const reef_fe &w2) {
        std::vector<reef_affine> out(n / 2);
        check(reef_fold(CURVE, gens, n / 2, REEF_HOST, &w1, &w2, out.data()), "reef_fold");
        return out;
    }

    // Cross terms (L, R) of IPA round k = w1s.size() over THESE generators (no fold): a = a_lo || a_hi.
    // CE::commit over the generators w1s.size() folds away from this key, slice [off, off + len), without folding them
    // (reef_msm_folded): what a CommitmentGens that records its folds returns from commit / split_at().commit
    reef_jacobian commit_folded(const reef_fe *v, size_t len, size_t off, const std::vector<reef_fe> &w1s, const std::vector<reef_fe> &w2s,
                                int v_loc = REEF_HOST) const {
        if (w1s.size() != w2s.size()) throw std::invalid_argument("challenge vectors differ in length");
        reef_jacobian out;
        check(reef_msm_folded(ctx_, v, len, off, v_loc, true, w1s.data(), w2s.data(), w1s.size(), &out, REEF_HOST), "reef_msm_folded");
        return out;
    }
    std::pair<reef_jacobian, reef_jacobian> ipa_cross_terms(const reef_fe *a, size_t n_k, const std::vector<reef_fe> &w1s,
                                                            const std::vector<reef_fe> &w2s, int a_loc = REEF_HOST) const {
        if (w1s.size() != w2s.size()) throw std::logic_error("one (w1, w2) pair per round");
        std::pair<reef_jacobian, reef_jacobian> lr;
        check(reef_ipa_cross_terms(ctx_, a, n_k, a_loc, true, w1s.data(), w2s.data(), w1s.size(), &lr.first, &lr.second), "reef_ipa_cross_terms");
        return lr;
    }

  private:
    reef_msm_ctx *ctx_ = nullptr;
    size_t n_;
    reef_affine h_ = {};
    bool has_h_;
};

// The same commitment key kept on several GPUs of ONE process -- Reef's prover is one process (src/backend/main.rs:82) -- through the
// library's device groups: CE::commit is split by Pippenger window over the devices (the 96-byte partial sums are exchanged and added
// inside the library), the rows of HyraxPC::commit are dealt out whole.  `devices` may repeat an ordinal.
template <int CURVE> class CommitmentGensOnDevices {
  public:
    // scalars: how the caller's host scalars reach the members of a windows group -- every member uploads them over its own PCIe link (REEF_SCALARS_EACH),
    // or one upload to devices[0] and peer copies from there (REEF_SCALARS_FANOUT); which wins depends on the host's topology (reef_msm.h section 5)
    CommitmentGensOnDevices(const reef_affine *gens, size_t n, const std::vector<int> &devices, const reef_affine *h = nullptr, uint32_t split = REEF_SPLIT_WINDOWS,
                            uint32_t scalars = REEF_SCALARS_EACH)
        : n_(n), has_h_(h != nullptr) {
        reef_msm_opts o = {};
        o.bucket_groups = 1;
        reef_msm_group_opts g = {};
        g.split = split;
        g.scalars = scalars;
        check(reef_msm_group_create(&grp_, CURVE, gens, n, REEF_HOST, &o, devices.data(), devices.size(), &g), "reef_msm_group_create");
        if (h) h_ = *h;
    }
    ~CommitmentGensOnDevices() { reef_msm_group_destroy(grp_); }
    CommitmentGensOnDevices(const CommitmentGensOnDevices &) = delete;
    CommitmentGensOnDevices &operator=(const CommitmentGensOnDevices &) = delete;
    size_t len() const { return n_; }
    // where the last commit's time went (distribution of the scalars, every member's stream, the combine): the first run on a real node explains itself
    void enable_timing(bool on) const { check(reef_msm_group_enable_timing(grp_, on ? 1 : 0), "reef_msm_group_enable_timing"); }
    reef_msm_group_timing last_timing() const {
        reef_msm_group_timing t;
        check(reef_msm_group_last_timing(grp_, &t), "reef_msm_group_last_timing");
        return t;
    }

    reef_jacobian commit(const reef_fe *v, size_t n, const reef_fe *blind = nullptr) const {
        reef_jacobian out;
        if (blind) {
            if (!has_h_) throw std::logic_error("commit with a blind needs the blinding generator");
            check(reef_msm_group_rows(grp_, v, 1, n, REEF_HOST, true, 0, blind, &h_, &out), "reef_msm_group_rows");
        } else {
            check(reef_msm_group_msm(grp_, v, n, REEF_HOST, true, &out), "reef_msm_group_msm");
        }
        return out;
    }
    // HyraxPC::commit(&poly) over the devices: rows x row_len field elements / one-byte symbols, one commitment per row
    std::vector<reef_jacobian> commit_rows(const reef_fe *poly, size_t rows, size_t row_len, const reef_fe *blinds, uint32_t max_scalar_bits = 0) const {
        std::vector<reef_jacobian> out(rows);
        check(reef_msm_group_rows(grp_, poly, rows, row_len, REEF_HOST, true, max_scalar_bits, blinds, blinds ? &h_ : nullptr, out.data()), "reef_msm_group_rows");
        return out;
    }
    std::vector<reef_jacobian> commit_symbols(const uint8_t *symbols, size_t rows, size_t row_len, uint32_t symbol_bits, const reef_fe *blinds) const {
        std::vector<reef_jacobian> out(rows);
        check(reef_msm_group_rows_symbols(grp_, symbols, rows, row_len, REEF_HOST, symbol_bits, blinds, blinds ? &h_ : nullptr, true, out.data()),
              "reef_msm_group_rows_symbols");
        return out;
    }

  private:
    reef_msm_group *grp_ = nullptr;
    size_t n_;
    reef_affine h_ = {};
    bool has_h_;
};

// nova-snark (fork) HyraxPC as Reef builds it at commitment.rs:182-185: 2^right row generators and the
// blinding generator; the polynomial's 2^l evaluations are viewed as a 2^left x 2^right matrix.
template <int CURVE> class HyraxPC {
  public:
    explicit HyraxPC(const CommitmentGens<CURVE> &gens_v) : gens_(gens_v) {
        if (!gens_v.h()) throw std::logic_error("HyraxPC needs a blinding generator");
    }
    static std::pair<size_t, size_t> compute_factored_lens(size_t num_vars) { return {num_vars / 2, num_vars - num_vars / 2}; }

    // HyraxPC::commit(&poly): one commitment per matrix row, sum_j Z[i,j] G_j + blinds[i] h.
    std::vector<reef_jacobian> commit(const reef_fe *poly, size_t num_vars, const reef_fe *blinds, uint32_t max_scalar_bits = 0) const {
        const auto lr = compute_factored_lens(num_vars);
        const size_t rows = (size_t)1 << lr.first, row_len = (size_t)1 << lr.second;
        if (row_len > gens_.len()) throw std::logic_error("not enough row generators");
        std::vector<reef_jacobian> out(rows);
        check(reef_msm_rows(gens_.handle(), poly, rows, row_len, REEF_HOST, true, max_scalar_bits, blinds, gens_.h(), out.data(), REEF_HOST),
              "reef_msm_rows");
        return out;
    }
    // The same from the document symbols themselves (one byte each, < 2^symbol_bits; framework.rs:978-1011).
    std::vector<reef_jacobian> commit_symbols(const uint8_t *symbols, size_t num_vars, uint32_t symbol_bits, const reef_fe *blinds,
                                              int symbols_loc = REEF_HOST) const {
        const auto lr = compute_factored_lens(num_vars);
        const size_t rows = (size_t)1 << lr.first, row_len = (size_t)1 << lr.second;
        if (row_len > gens_.len()) throw std::logic_error("not enough row generators");
        std::vector<reef_jacobian> out(rows);
        check(reef_msm_rows_symbols(gens_.handle(), symbols, rows, row_len, symbols_loc, symbol_bits, blinds, blinds ? gens_.h() : nullptr, true,
                                    out.data(), REEF_HOST),
              "reef_msm_rows_symbols");
        return out;
    }
    // First step of prove_eval: LZ = L^T Z and eval = <LZ, R> for (L, R) = eq-evaluations of the two halves of `point`.
    struct BoundRows {
        std::vector<reef_fe> lz;
        reef_fe eval;
    };
    BoundRows bind_rows(const void *z, size_t n, int elem_bytes, int z_loc, const reef_fe *point, size_t num_vars, bool is_mont = true) const {
        const auto lr = compute_factored_lens(num_vars);
        BoundRows b;
        b.lz.resize((size_t)1 << lr.second);
        check(reef_mle_bound_rows(CURVE, z, n, elem_bytes, z_loc, is_mont, point, num_vars, lr.first, b.lz.data(), REEF_HOST, &b.eval),
              "reef_mle_bound_rows");
        return b;
    }

  private:
    const CommitmentGens<CURVE> &gens_;
};

// The nlookup sum-check of witness generation (r1cs.rs:2318-2385) over the scalar field of Pallas: resident
// (table, eq) pair, one call per half of linear_mle_product; the Poseidon challenge stays with the caller.
class SumCheck {
  public:
    SumCheck(size_t ell) : ell_(ell) { check(reef_sc_create(&sc_, REEF_PALLAS, (size_t)1 << ell), "reef_sc_create"); }
    ~SumCheck() { reef_sc_destroy(sc_); }
    SumCheck(const SumCheck &) = delete;
    SumCheck &operator=(const SumCheck &) = delete;

    void set_table(const reef_fe *values, size_t n, int loc = REEF_HOST) { have_next_ = false; check(reef_sc_set_table(sc_, 0, values, n, loc), "reef_sc_set_table"); }
    void start_step() { have_next_ = false; check(reef_sc_reset_table(sc_), "reef_sc_reset_table"); }   // every folding step starts from the unfolded table
    void gen_eq_table(const std::vector<reef_fe> &rs, const std::vector<uint32_t> &qs, const std::vector<reef_fe> &last_q) {
        if (rs.size() != qs.size() + 1 || last_q.size() != ell_) throw std::logic_error("gen_eq_table: |rs| = |qs| + 1, |last_q| = ell");
        have_next_ = false;
        check(reef_sc_gen_eq_table(sc_, rs.data(), qs.data(), qs.size(), last_q.data(), ell_), "reef_sc_gen_eq_table");
    }
    // round i in 1..ell: (xsq, x, con)
    std::array<reef_fe, 3> round_coeffs(size_t i) {
        std::array<reef_fe, 3> g;
        check(reef_sc_round_coeffs(sc_, (size_t)1 << (ell_ - i), g.data()), "reef_sc_round_coeffs");
        return g;
    }
    void fold(size_t i, const reef_fe &r) { check(reef_sc_fold(sc_, (size_t)1 << (ell_ - i), &r), "reef_sc_fold"); }
    // fold of round i and the coefficients of round i + 1 in one pass (i < ell)
    std::array<reef_fe, 3> fold_and_next_coeffs(size_t i, const reef_fe &r) {
        std::array<reef_fe, 3> g;
        check(reef_sc_fold_and_next_coeffs(sc_, (size_t)1 << (ell_ - i), &r, g.data()), "reef_sc_fold_and_next_coeffs");
        return g;
    }
    // The whole of the reference's linear_mle_product(table_t, table_eq, ell, i, sponge) (r1cs_helper.rs:441-506) on the resident tables: the round's
    // sums, absorb (con, x, xsq) in that order, squeeze the challenge, fold both tables; returns {r_i, xsq, x, con} like the reference.  `transcript`
    // is the caller's sponge: given {con, x, xsq} (canonical) it returns the challenge (neptune's SpongeAPI absorb(3) + squeeze(1) on the Rust side).
    // Driven round after round as r1cs.rs:2318-2385 does, each round is one pass over the tables (the fold of round i yields round i + 1's sums).
    template <class Transcript> std::array<reef_fe, 4> linear_mle_product(size_t i, Transcript &&transcript) {
        const std::array<reef_fe, 3> g = (have_next_ && next_round_ == i) ? next_ : round_coeffs(i);      // {xsq, x, con}
        const std::array<reef_fe, 3> query = {g[2], g[1], g[0]};
        const reef_fe r = transcript(query);
        have_next_ = false;
        if (i < ell_) {
            next_ = fold_and_next_coeffs(i, r);
            next_round_ = i + 1;
            have_next_ = true;
        } else {
            fold(i, r);
        }
        return {r, g[0], g[1], g[2]};
    }
    reef_fe final_value() {   // prover_mle_partial_eval(table, sc_rs) after the last fold (r1cs.rs:2379-2385)
        reef_fe v;
        check(reef_sc_read(sc_, 0, 1, &v), "reef_sc_read");
        return v;
    }

  private:
    reef_sc_ctx *sc_ = nullptr;
    size_t ell_;
    std::array<reef_fe, 3> next_ = {};     // sums of the round after the last linear_mle_product, if that call produced them
    size_t next_round_ = 0;
    bool have_next_ = false;
};

// The reference's MerkleCommitment<F> (src/backend/merkle_tree.rs:11-16): `commitment` (the root), `tree` (the levels, the leaves' parents first)
// and `doc`.  The tree is built on the GPU -- on one device, or in blocks over `devices` (reef_merkle_commit_devices) --; path_wits / make_wits
// (:116-190) are look-ups in it on the host, as in the reference.  The Poseidon constants are the caller's (neptune's, on the Rust side).
struct MerkleWit {
    bool l_or_r;          // true: the node on the path is the LEFT child
    bool has_idx;         // leaf level only (the reference's Option): opposite_idx is the sibling's document index (0 when it is missing)
    uint64_t opposite_idx;
    reef_fe opposite;     // the sibling's value, canonical (zero when it is missing)
};
template <int CURVE> class MerkleCommitment {
  public:
    MerkleCommitment(const std::vector<uint32_t> &document, const reef_poseidon_params &pc, bool pc_is_mont = false, const std::vector<int> &devices = {})
        : doc(document) {
        if (doc.empty()) throw std::invalid_argument("MerkleCommitment::new: empty document");     // the reference indexes an empty level: a panic
        std::vector<reef_fe> flat(reef_merkle_nodes(doc.size()));
        if (devices.empty())
            check(reef_merkle_commit(CURVE, &pc, doc.data(), doc.size(), REEF_HOST, pc_is_mont, flat.data(), REEF_HOST, &commitment), "reef_merkle_commit");
        else
            check(reef_merkle_commit_devices(CURVE, &pc, doc.data(), doc.size(), pc_is_mont, devices.data(), devices.size(), flat.data(), &commitment, nullptr),
                  "reef_merkle_commit_devices");
        size_t m = (doc.size() + 1) / 2, off = 0;
        for (;;) {
            tree.emplace_back(flat.begin() + off, flat.begin() + off + m);
            off += m;
            if (m <= 1) break;
            m = (m + 1) / 2;
        }
    }
    // merkle_tree.rs:128-190
    std::vector<MerkleWit> path_wits(size_t idx) const {
        if (idx >= doc.size()) throw std::out_of_range("path_wits: idx < doc.len()");
        const reef_fe zero = {};
        auto sym = [](uint32_t v) { return reef_fe{{v, 0, 0, 0}}; };
        std::vector<MerkleWit> w;
        if (idx % 2 == 0) {
            if (idx + 1 >= doc.size()) w.push_back({true, true, 0, zero});
            else w.push_back({true, true, idx + 1, sym(doc[idx + 1])});
        } else {
            w.push_back({false, true, idx - 1, sym(doc[idx - 1])});
        }
        size_t quo = idx / 2;
        for (size_t h = 0; h + 1 < tree.size(); ++h) {
            if (quo % 2 == 0) w.push_back({true, false, 0, quo + 1 >= tree[h].size() ? zero : tree[h][quo + 1]});
            else w.push_back({false, false, 0, tree[h][quo - 1]});
            quo /= 2;
        }
        return w;
    }
    // merkle_tree.rs:116-126
    std::vector<std::vector<MerkleWit>> make_wits(const std::vector<size_t> &m_lookups) const {
        std::vector<std::vector<MerkleWit>> wits;
        for (size_t q : m_lookups) wits.push_back(path_wits(q));
        return wits;
    }

    reef_fe commitment = {};
    std::vector<std::vector<reef_fe>> tree;
    std::vector<uint32_t> doc;
};

}  // namespace reef_provider
