// sc_stress -- the one-launch sum-check rounds under load, through the C ABI only.
//
// A multi-block one-launch round (reef_amd/csrc/sumcheck_kernels.inc: sc_round_epilogue) hands the blocks' sums to the last block
// through device-scope atomics; by default the ticket is an acq_rel read-modify-write (REEF_SC_FENCE=2); with REEF_SC_FENCE=0 nothing but the return of those atomics orders them before the ticket.
// Functional tests do not find a once-in-10^6 ordering bug, so this program repeats ONE folding step (reef_sc_gen_eq_table, then
// every round fused: src/backend/r1cs_helper.rs:441-544) thousands of times with the same inputs and compares every coefficient
// triple with the transcript of the two-launch form (REEF_SC_ONE_LAUNCH=0: a second kernel adds the blocks' sums after a kernel
// boundary), while -- with `load` -- two other caller threads keep the L2s and the fabric busy: 2^18-point MSMs (k_accum0's
// 64-byte gathers) and fused rounds over a 2^24-entry table (HBM streaming).  The grid shapes come from the library's own switches in
// the ENVIRONMENT of the process (REEF_SC_BLOCKS, REEF_SC_ITEMS, REEF_SC_ONE_LAUNCH_MAX, REEF_SC_SPLIT_MAX, REEF_SC_RANK1_MIN_POW,
// REEF_SC_FENCE, ...): tests/test_gpu_sumcheck.py sets them.  Prints one JSON line; exit code 1 on a mismatch.
//
// usage: sc_stress <ell> <steps> [load] [print]      (print: the reference transcript in hex, for the oracle)
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "reef_msm.h"

#define CK(x)                                                                                   \
    do {                                                                                        \
        reef_status s_ = (x);                                                                   \
        if (s_ != REEF_OK) { fprintf(stderr, "sc_stress: %s failed: %s\n", #x, reef_last_error()); exit(3); } \
    } while (0)

struct Inputs {
    int ell, nq;
    std::vector<reef_fe> rs, last_q, chal;
    std::vector<uint32_t> qs;
};
static Inputs make_inputs(int ell) {
    Inputs in;
    in.ell = ell;
    in.nq = 9;
    for (int i = 0; i <= in.nq; ++i) in.rs.push_back(reef_fe{{0x9e3779b97f4a7c15ULL * (i + 1), 0x1234ULL + i, 0x55aaULL * i, 0x0123456789abcdefULL}});
    for (int i = 0; i < in.nq; ++i) in.qs.push_back((uint32_t)((0x2545F4914F6CDD1DULL * (i + 7)) >> (64 - ell)));
    in.qs[1] = in.qs[0];                                // a repeated lookup index
    for (int j = 0; j < ell; ++j) in.last_q.push_back(reef_fe{{0xabcdef12345ULL + j, 0x77ULL * j, 0xfeedULL, 0x0fedcba987654321ULL}});
    for (int i = 1; i <= ell; ++i) in.chal.push_back(reef_fe{{0x5851f42d4c957f2dULL + i, 0x14057b7ef767814fULL, 0x0123456789abcdefULL, 0x0fedcba987654321ULL}});
    return in;
}
// one folding step; the coefficients of every round and the final table value are appended to `t`
static void run_step(reef_sc_ctx *sc, const Inputs &in, std::vector<reef_fe> &t) {
    CK(reef_sc_reset_table(sc));
    CK(reef_sc_gen_eq_table(sc, in.rs.data(), in.qs.data(), in.nq, in.last_q.data(), in.ell));
    reef_fe g[3];
    CK(reef_sc_round_coeffs(sc, (size_t)1 << (in.ell - 1), g));
    t.insert(t.end(), g, g + 3);
    for (int i = 1; i <= in.ell; ++i) {
        const size_t pow = (size_t)1 << (in.ell - i);
        if (pow >= 2) {
            CK(reef_sc_fold_and_next_coeffs(sc, pow, &in.chal[i - 1], g));
            t.insert(t.end(), g, g + 3);
        } else {
            CK(reef_sc_fold(sc, pow, &in.chal[i - 1]));
        }
    }
    reef_fe v;
    CK(reef_sc_read(sc, 0, 1, &v));
    t.push_back(v);
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: sc_stress <ell> <steps> [load] [print]\n"); return 2; }
    const int ell = atoi(argv[1]);
    const long steps = atol(argv[2]);
    bool load = false, print = false;
    for (int i = 3; i < argc; ++i) { load = load || !strcmp(argv[i], "load"); print = print || !strcmp(argv[i], "print"); }
    if (ell < 2 || ell > 26 || steps < 1) { fprintf(stderr, "sc_stress: ell in 2..26, steps >= 1\n"); return 2; }
    {
        reef_runtime_opts ro = {};
        ro.hw_queues = 8;
        (void)reef_runtime_init(&ro, nullptr);
    }
    if (reef_device_count() < 1) { fprintf(stderr, "sc_stress: no GPU\n"); return 3; }
    const size_t len = (size_t)1 << ell;
    const Inputs in = make_inputs(ell);
    reef_sc_ctx *sc = nullptr;
    CK(reef_sc_create(&sc, REEF_PALLAS, len));
    {
        reef_fe *d_tab = (reef_fe *)reef_device_alloc(len * sizeof(reef_fe));
        if (!d_tab) { fprintf(stderr, "sc_stress: alloc\n"); return 3; }
        CK(reef_gen_scalars(REEF_PALLAS, 0x7AB1E + ell, 0, 0, len, false, d_tab, REEF_DEVICE));   // full-width entries: dense rows
        CK(reef_sc_set_table(sc, 0, d_tab, len, REEF_DEVICE));
        reef_device_free(d_tab);
    }
    // the reference transcript: the two-launch form of the same step, EQ written out (no rank-one rounds)
    const char *keep_r1 = getenv("REEF_SC_RANK1");
    const std::string keep_r1s = keep_r1 ? keep_r1 : "";
    setenv("REEF_SC_ONE_LAUNCH", "0", 1);
    setenv("REEF_SC_RANK1", "0", 1);
    std::vector<reef_fe> ref;
    run_step(sc, in, ref);
    setenv("REEF_SC_ONE_LAUNCH", "1", 1);
    if (keep_r1) setenv("REEF_SC_RANK1", keep_r1s.c_str(), 1);
    else unsetenv("REEF_SC_RANK1");

    std::atomic<bool> stop{false};
    std::atomic<long> load_msms{0}, load_rounds{0};
    std::vector<std::thread> bg;
    if (load) {
        bg.emplace_back([&] {                           // k_accum0: 64-byte gathers from sixteen 4 MiB tables... of a 2^18-point key
            const size_t n = (size_t)1 << 18;
            reef_affine *d_b = (reef_affine *)reef_device_alloc(n * sizeof(reef_affine));
            reef_fe *d_s = (reef_fe *)reef_device_alloc(n * sizeof(reef_fe));
            reef_jacobian *d_o = (reef_jacobian *)reef_device_alloc(sizeof(reef_jacobian));
            CK(reef_gen_bases(REEF_VESTA, 99, 3, n, d_b, REEF_DEVICE));
            CK(reef_gen_scalars(REEF_VESTA, 5, 0, 0, n, true, d_s, REEF_DEVICE));
            reef_msm_opts o = {};
            o.bucket_groups = 1;
            o.device = -1;
            reef_msm_ctx *k = nullptr;
            CK(reef_msm_ctx_create(&k, REEF_VESTA, d_b, n, REEF_DEVICE, &o));
            while (!stop.load()) {
                for (int i = 0; i < 4; ++i) CK(reef_msm(k, d_s, n, REEF_DEVICE, true, d_o, REEF_DEVICE));
                CK(reef_msm_ctx_sync(k));
                load_msms += 4;
            }
            reef_msm_ctx_destroy(k);
            reef_device_free(d_b); reef_device_free(d_s); reef_device_free(d_o);
        });
        bg.emplace_back([&] {                           // HBM streaming: fused rounds over a 2^24-entry table, again and again
            const int bell = 24;
            const size_t blen = (size_t)1 << bell;
            const Inputs bin = make_inputs(bell);
            reef_sc_ctx *b = nullptr;
            CK(reef_sc_create(&b, REEF_PALLAS, blen));
            reef_fe *d_tab = (reef_fe *)reef_device_alloc(blen * sizeof(reef_fe));
            CK(reef_gen_scalars(REEF_PALLAS, 0xB16, 0, 0, blen, false, d_tab, REEF_DEVICE));
            CK(reef_sc_set_table(b, 0, d_tab, blen, REEF_DEVICE));
            reef_device_free(d_tab);
            reef_fe g[3];
            while (!stop.load()) {
                CK(reef_sc_reset_table(b));
                CK(reef_sc_gen_eq_table(b, bin.rs.data(), bin.qs.data(), bin.nq, bin.last_q.data(), bell));
                CK(reef_sc_round_coeffs(b, blen / 2, g));
                for (int i = 1; i <= 4; ++i) CK(reef_sc_fold_and_next_coeffs(b, blen >> i, &bin.chal[i - 1], g));   // the four large rounds
                load_rounds += 5;
            }
            reef_sc_destroy(b);
        });
    }
    long mismatched_steps = 0, mismatched_values = 0, first_bad_step = -1, first_bad_index = -1;
    std::vector<reef_fe> got;
    const auto t0 = std::chrono::steady_clock::now();
    for (long s = 0; s < steps; ++s) {
        got.clear();
        run_step(sc, in, got);
        if (got.size() != ref.size() || memcmp(got.data(), ref.data(), ref.size() * sizeof(reef_fe)) != 0) {
            ++mismatched_steps;
            for (size_t i = 0; i < ref.size() && i < got.size(); ++i)
                if (memcmp(&got[i], &ref[i], sizeof(reef_fe)) != 0) {
                    ++mismatched_values;
                    if (first_bad_step < 0) { first_bad_step = s; first_bad_index = (long)i; }
                }
        }
    }
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop.store(true);
    for (auto &t : bg) t.join();
    reef_sc_destroy(sc);
    auto env = [](const char *k) { const char *e = getenv(k); return std::string(e ? e : "default"); };
    printf("{\"ell\": %d, \"steps\": %ld, \"rounds_compared\": %ld, \"mismatched_steps\": %ld, \"mismatched_values\": %ld, \"first_bad_step\": %ld, \"first_bad_index\": %ld, "
           "\"seconds\": %.3f, \"load\": %s, \"load_msms\": %ld, \"load_streaming_rounds\": %ld, \"REEF_SC_FENCE\": \"%s\", \"REEF_SC_BLOCKS\": \"%s\", \"REEF_SC_ITEMS\": \"%s\", "
           "\"REEF_SC_ONE_LAUNCH_MAX\": \"%s\", \"REEF_SC_SPLIT_MAX\": \"%s\", \"REEF_SC_SPLIT_BLOCKS\": \"%s\", \"REEF_SC_RANK1_MIN_POW\": \"%s\", \"reference\": \"two-launch form (REEF_SC_ONE_LAUNCH=0, REEF_SC_RANK1=0) of the same step\"",
           ell, steps, steps * (long)ell, mismatched_steps, mismatched_values, first_bad_step, first_bad_index, secs, load ? "true" : "false", load_msms.load(), load_rounds.load(),
           env("REEF_SC_FENCE").c_str(), env("REEF_SC_BLOCKS").c_str(), env("REEF_SC_ITEMS").c_str(), env("REEF_SC_ONE_LAUNCH_MAX").c_str(), env("REEF_SC_SPLIT_MAX").c_str(),
           env("REEF_SC_SPLIT_BLOCKS").c_str(), env("REEF_SC_RANK1_MIN_POW").c_str());
    if (print) {
        printf(", \"transcript\": [");
        for (size_t i = 0; i < ref.size(); ++i) printf("%s\"%016llx%016llx%016llx%016llx\"", i ? ", " : "", (unsigned long long)ref[i].l[3], (unsigned long long)ref[i].l[2],
                                                       (unsigned long long)ref[i].l[1], (unsigned long long)ref[i].l[0]);
        printf("]");
    }
    printf("}\n");
    return mismatched_steps ? 1 : 0;
}
