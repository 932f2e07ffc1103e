// Exercises the C++ provider mirror (reef_provider.hpp) on seeded inputs and prints one JSON object
// of results (32-byte compressed points / canonical field elements, hex).  tests/test_gpu_provider.py
// recomputes every entry with the oracle from the same seeds and compares bit for bit.
// Build: g++ -O2 -std=c++17 provider_selftest.cpp -I../../../include -L../../_lib -lreef_msm -o provider_selftest
#include <cstdio>
#include <cstring>
#include <string>

#include "reef_provider.hpp"
#include "replay_standins.h"   // stand-in hash-to-curve / Poseidon constants (generated from the oracle's; NOT pasta_curves' / neptune's)

using namespace reef_provider;

static std::string hex(const uint8_t *p, size_t n) {
    static const char *d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; ++i) { s += d[p[i] >> 4]; s += d[p[i] & 15]; }
    return s;
}
template <int C> static std::string chex(const reef_jacobian &p) { const Compressed c = compress<C>(p); return hex(c.data(), 32); }
static std::string fhex(const reef_fe &f) { return hex((const uint8_t *)&f, 32); }
static reef_fe small(uint64_t v) { return reef_fe{{v, 0, 0, 0}}; }

int main() {
    try {
        if (reef_device_count() < 1) { fprintf(stderr, "no GPU: %s\n", reef_last_error()); return 3; }
        std::string out = "{";
        // ---- CE::commit with a blind, Pallas, 1000 generators (k0 = 77, d = 13), blinding generator 0xB11D * G
        const size_t n = 1000;
        std::vector<reef_affine> gens(n);
        reef_affine h;
        check(reef_gen_bases(REEF_PALLAS, 77, 13, n, gens.data(), REEF_HOST), "gen_bases");
        check(reef_gen_bases(REEF_PALLAS, 0xB11D, 1, 1, &h, REEF_HOST), "gen_bases");
        std::vector<reef_fe> v(n), blind(1);
        check(reef_gen_scalars(REEF_PALLAS, 4242, 1, 0, n, true, v.data(), REEF_HOST), "gen_scalars");
        check(reef_gen_scalars(REEF_PALLAS, 4243, 0, 0, 1, true, blind.data(), REEF_HOST), "gen_scalars");
        CommitmentGens<REEF_PALLAS> ck(gens.data(), n, REEF_HOST, &h);
        out += "\"commit_blind\": \"" + chex<REEF_PALLAS>(ck.commit(v.data(), n, blind.data())) + "\", ";
        out += "\"commit\": \"" + chex<REEF_PALLAS>(ck.commit(v.data(), n)) + "\", ";
        // ---- HyraxPC::commit of a 9-variable polynomial (16 rows x 32), ASCII-like symbols, from field elements and from bytes
        const size_t vars = 9, rows = 16, cols = 32;
        std::vector<reef_affine> rg(cols);
        check(reef_gen_bases(REEF_PALLAS, 3, 7, cols, rg.data(), REEF_HOST), "gen_bases");
        CommitmentGens<REEF_PALLAS> gv(rg.data(), cols, REEF_HOST, &h);
        HyraxPC<REEF_PALLAS> pc(gv);
        std::vector<reef_fe> z(rows * cols), zc(rows * cols), bl(rows);
        check(reef_gen_scalars(REEF_PALLAS, 11, 2, 131, rows * cols, true, z.data(), REEF_HOST), "gen_scalars");
        check(reef_gen_scalars(REEF_PALLAS, 11, 2, 131, rows * cols, false, zc.data(), REEF_HOST), "gen_scalars");
        check(reef_gen_scalars(REEF_PALLAS, 12, 0, 0, rows, true, bl.data(), REEF_HOST), "gen_scalars");
        std::vector<uint8_t> sym(rows * cols);
        for (size_t i = 0; i < sym.size(); ++i) sym[i] = (uint8_t)zc[i].l[0];
        const auto c1 = pc.commit(z.data(), vars, bl.data());
        const auto c2 = pc.commit_symbols(sym.data(), vars, 8, bl.data());
        out += "\"hyrax\": [";
        for (size_t i = 0; i < rows; ++i) out += std::string(i ? ", " : "") + "\"" + chex<REEF_PALLAS>(c1[i]) + "\"";
        out += "], \"hyrax_symbols\": [";
        for (size_t i = 0; i < rows; ++i) out += std::string(i ? ", " : "") + "\"" + chex<REEF_PALLAS>(c2[i]) + "\"";
        out += "], ";
        // ---- the same commitments with the keys on three members of a device group (one process; the ordinals repeat device 0)
        {
            CommitmentGensOnDevices<REEF_PALLAS> ckd(gens.data(), n, {0, 0, 0}, &h);
            out += "\"group_commit_blind\": \"" + chex<REEF_PALLAS>(ckd.commit(v.data(), n, blind.data())) + "\", ";
            out += "\"group_commit\": \"" + chex<REEF_PALLAS>(ckd.commit(v.data(), n)) + "\", ";
            {   // the same key with the host scalars fanned out from devices[0], and the call itemised (round 6)
                CommitmentGensOnDevices<REEF_PALLAS> ckf(gens.data(), n, {0, 0, 0}, &h, REEF_SPLIT_WINDOWS, REEF_SCALARS_FANOUT);
                ckf.enable_timing(true);
                out += "\"group_commit_fanout\": \"" + chex<REEF_PALLAS>(ckf.commit(v.data(), n)) + "\", ";
                const reef_msm_group_timing t = ckf.last_timing();
                out += std::string("\"group_timing_consistent\": ") + (t.members == 3 && t.total_ms > 0 && t.distribute_ms <= t.members_done_ms && t.members_done_ms <= t.total_ms ? "true" : "false") + ", ";
            }
            CommitmentGensOnDevices<REEF_PALLAS> gvd(rg.data(), cols, {0, 0, 0}, &h);
            const auto d1 = gvd.commit_rows(z.data(), rows, cols, bl.data());
            const auto d2 = gvd.commit_symbols(sym.data(), rows, cols, 8, bl.data());
            out += "\"group_hyrax\": [";
            for (size_t i = 0; i < rows; ++i) out += std::string(i ? ", " : "") + "\"" + chex<REEF_PALLAS>(d1[i]) + "\"";
            out += "], \"group_hyrax_symbols\": [";
            for (size_t i = 0; i < rows; ++i) out += std::string(i ? ", " : "") + "\"" + chex<REEF_PALLAS>(d2[i]) + "\"";
            out += "], ";
        }
        // ---- row binding of prove_eval at a fixed point (canonical integers)
        std::vector<reef_fe> point(vars);
        for (size_t j = 0; j < vars; ++j) point[j] = reef_fe{{0x1f83d9abfb41bd6bULL + j, 0x5be0cd19137e2179ULL, 0x3c6ef372fe94f82bULL, 0x0a54ff53a5f1d36fULL}};
        const auto br = pc.bind_rows(sym.data(), sym.size(), 1, REEF_HOST, point.data(), vars, false);
        out += "\"bind_eval\": \"" + fhex(br.eval) + "\", \"bind_lz0\": \"" + fhex(br.lz[0]) + "\", ";
        // ---- IPA cross terms over 256 Vesta generators (k0 = 21, d = 4), rounds 0 and 2
        const size_t m = 256;
        std::vector<reef_affine> ig(m);
        check(reef_gen_bases(REEF_VESTA, 21, 4, m, ig.data(), REEF_HOST), "gen_bases");
        CommitmentGens<REEF_VESTA> ik(ig.data(), m);
        std::vector<reef_fe> a(m);
        check(reef_gen_scalars(REEF_VESTA, 3, 0, 0, m, true, a.data(), REEF_HOST), "gen_scalars");
        std::vector<reef_fe> w1s, w2s;
        auto lr = ik.ipa_cross_terms(a.data(), m, w1s, w2s);
        out += "\"ipa_l0\": \"" + chex<REEF_VESTA>(lr.first) + "\", \"ipa_r0\": \"" + chex<REEF_VESTA>(lr.second) + "\", ";
        w1s = {small(5), small(7)};
        w2s = {small(11), small(13)};
        lr = ik.ipa_cross_terms(a.data(), m / 4, w1s, w2s);
        out += "\"ipa_l2\": \"" + chex<REEF_VESTA>(lr.first) + "\", \"ipa_r2\": \"" + chex<REEF_VESTA>(lr.second) + "\", ";
        // ---- the reference's mle_linear_basic inputs (r1cs.rs:2411-2515) through SumCheck
        SumCheck sc(3);
        const uint64_t evals[8] = {2, 3, 5, 7, 9, 13, 17, 19};
        std::vector<reef_fe> tab(8);
        for (int i = 0; i < 8; ++i) tab[i] = small(evals[i]);
        sc.set_table(tab.data(), 8);
        sc.start_step();
        sc.gen_eq_table({small(3), small(9), small(27), small(81)}, {2, 1, 7}, {small(5), small(3), small(2)});   // last_q reversed, as Reef passes it
        const reef_fe rs[3] = {small(5), small(1000003), small(0x1234567890ABCDEFULL)};
        out += "\"sumcheck\": [";
        auto g = sc.round_coeffs(1);
        for (size_t i = 1; i <= 3; ++i) {
            out += std::string(i > 1 ? ", " : "") + "[\"" + fhex(g[0]) + "\", \"" + fhex(g[1]) + "\", \"" + fhex(g[2]) + "\"]";
            if (i < 3) g = sc.fold_and_next_coeffs(i, rs[i - 1]);
            else sc.fold(i, rs[i - 1]);
        }
        out += "], \"sumcheck_final\": \"" + fhex(sc.final_value()) + "\", ";
        // ---- the same step through linear_mle_product (the reference's one-call round), with a transcript that returns fixed challenges and records what it absorbed
        {
            sc.start_step();
            sc.gen_eq_table({small(3), small(9), small(27), small(81)}, {2, 1, 7}, {small(5), small(3), small(2)});
            size_t round = 0;
            std::string absorbed = "[";
            auto transcript = [&](const std::array<reef_fe, 3> &q) {
                absorbed += std::string(round ? ", " : "") + "[\"" + fhex(q[0]) + "\", \"" + fhex(q[1]) + "\", \"" + fhex(q[2]) + "\"]";
                return rs[round++];
            };
            out += "\"lmp\": [";
            for (size_t i = 1; i <= 3; ++i) {
                const auto t4 = sc.linear_mle_product(i, transcript);
                out += std::string(i > 1 ? ", " : "") + "[\"" + fhex(t4[0]) + "\", \"" + fhex(t4[1]) + "\", \"" + fhex(t4[2]) + "\", \"" + fhex(t4[3]) + "\"]";
            }
            out += "], \"lmp_absorbed\": " + absorbed + "], \"lmp_final\": \"" + fhex(sc.final_value()) + "\", ";
        }
        // ---- CommitmentGens::new(label, n) / new_with_blinding_gen: the key derived from a label on the GPU (row N1), commitments over it
        {
            reef_keygen_params kp;
            kp.a = STANDIN_KEYGEN_0[0]; kp.b = STANDIN_KEYGEN_0[1]; kp.z = STANDIN_KEYGEN_0[2];
            memcpy(kp.iso, STANDIN_KEYGEN_0 + 3, 13 * sizeof(reef_fe));
            kp.dst = (const uint8_t *)STANDIN_KEYGEN_DST_0; kp.dst_len = (uint32_t)strlen(STANDIN_KEYGEN_DST_0); kp.little_endian = 0;
            const size_t ln = 300;
            CommitmentGens<REEF_PALLAS> lk("reef ck", ln, kp, &h);
            std::vector<reef_fe> lv(ln);
            check(reef_gen_scalars(REEF_PALLAS, 99, 1, 0, ln, true, lv.data(), REEF_HOST), "gen_scalars");
            out += "\"label_commit\": \"" + chex<REEF_PALLAS>(lk.commit(lv.data(), ln)) + "\", ";
            out += "\"label_commit_blind\": \"" + chex<REEF_PALLAS>(lk.commit(lv.data(), ln, blind.data())) + "\", ";
            const auto lg = CommitmentGens<REEF_PALLAS>::from_label("reef ck", 3, kp);
            out += "\"label_gens\": \"" + hex((const uint8_t *)lg.data(), 3 * sizeof(reef_affine)) + "\", ";
        }
        // ---- MerkleCommitment::new(&doc, &pc), path_wits (row N4; the reference's own document of make_mt, merkle_tree.rs:214, and a longer one in blocks)
        {
            reef_poseidon_params pp;
            pp.width = 5; pp.full_rounds = STANDIN_POSEIDON_RF; pp.partial_rounds = STANDIN_POSEIDON_RP; pp.reserved = 0;
            pp.round_constants = STANDIN_POSEIDON_RC; pp.mds = STANDIN_POSEIDON_MDS;
            pp.tag_leaf = STANDIN_POSEIDON_TAGS[0]; pp.tag_node = STANDIN_POSEIDON_TAGS[1];
            auto path_json = [&](const std::vector<MerkleWit> &w) {
                std::string j = "[";
                for (size_t i = 0; i < w.size(); ++i)
                    j += std::string(i ? ", " : "") + "[" + (w[i].l_or_r ? "true" : "false") + ", " + (w[i].has_idx ? std::to_string(w[i].opposite_idx) : "null") +
                         ", \"" + fhex(w[i].opposite) + "\"]";
                return j + "]";
            };
            MerkleCommitment<REEF_PALLAS> mt({2, 3, 4, 5, 6, 7, 8}, pp);
            out += "\"merkle_root\": \"" + fhex(mt.commitment) + "\", \"merkle_levels\": " + std::to_string(mt.tree.size()) + ", ";
            out += "\"merkle_path6\": " + path_json(mt.path_wits(6)) + ", \"merkle_path3\": " + path_json(mt.make_wits({0, 3})[1]) + ", ";
            std::vector<uint32_t> longdoc(1001);
            for (size_t i = 0; i < longdoc.size(); ++i) longdoc[i] = (uint32_t)((31 * i + 7) % 131);
            MerkleCommitment<REEF_PALLAS> mb(longdoc, pp, false, {0, 0, 0});
            out += "\"merkle_blocks_root\": \"" + fhex(mb.commitment) + "\", \"merkle_blocks_path500\": " + path_json(mb.path_wits(500)) + ", ";
            bool oob = false;
            try { mt.path_wits(7); } catch (const std::out_of_range &) { oob = true; }
            out += std::string("\"merkle_oob_throws\": ") + (oob ? "true" : "false") + ", ";
        }
        // ---- error behaviour: a failure is an exception carrying the library's message
        bool threw = false;
        try { ck.commit(v.data(), n + 1); } catch (const Error &e) { threw = e.status != REEF_OK; }
        out += std::string("\"error_throws\": ") + (threw ? "true" : "false") + "}";
        printf("%s\n", out.c_str());
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "provider_selftest: %s\n", e.what());
        return 1;
    }
}
