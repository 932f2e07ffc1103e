// seam_bench -- the pasta-msm drop-in symbols under the callers Reef really has: T threads inside mult_pippenger_pallas at once on ONE
// returning key (nova-snark's prover thread and rayon workers: src/backend/framework.rs:110, 668, 695), host buffers in, commitment
// back (PCIe-inclusive).  C++ so that no interpreter lock sits in the measurement.  Per (size, thread count): `calls` calls per thread,
// `reps` repetitions; reports the MEDIAN, minimum and maximum aggregate rate over the repetitions, the median per-call latency, and where
// the calls spent their host time (reef_key_cache_timing_get: nominate / enqueue incl. the staging of the pageable scalars / confirm
// the key's bytes / wait).  One line of JSON per (size, threads).
//
// usage: seam_bench [sizes=27790,65536] [threads=1,4,8] [calls=500] [reps=5]
//        seam_bench first=1 [sizes=4096,16384,65536] [gap_us=0,2000]     the first eight calls on a fresh key, one by one
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "reef_msm.h"

#define CK(x)                                                                                                    \
    do {                                                                                                         \
        reef_status s_ = (x);                                                                                    \
        if (s_ != REEF_OK) { fprintf(stderr, "seam_bench: %s failed: %s\n", #x, reef_last_error()); exit(3); }    \
    } while (0)

static std::vector<long> list_of(const char *s) {
    std::vector<long> v;
    while (*s) {
        v.push_back(atol(s));
        const char *c = strchr(s, ',');
        if (!c) break;
        s = c + 1;
    }
    return v;
}
using clk = std::chrono::steady_clock;

int main(int argc, char **argv) {
    std::vector<long> sizes = {3000, 27790, 65536, 262144}, threads = {1, 4, 8};
    long calls = 500, reps = 5, first = 0, cold_n = 27790, cold_warm = 0, host_ms = 300;
    std::string gen_path, cold_path;
    std::vector<long> gaps = {0};
    for (int i = 1; i < argc; ++i) {
        if (!strncmp(argv[i], "sizes=", 6)) sizes = list_of(argv[i] + 6);
        else if (!strncmp(argv[i], "threads=", 8)) threads = list_of(argv[i] + 8);
        else if (!strncmp(argv[i], "calls=", 6)) calls = atol(argv[i] + 6);
        else if (!strncmp(argv[i], "reps=", 5)) reps = atol(argv[i] + 5);
        else if (!strncmp(argv[i], "first=", 6)) first = atol(argv[i] + 6);
        else if (!strncmp(argv[i], "gen=", 4)) gen_path = argv[i] + 4;
        else if (!strncmp(argv[i], "cold=", 5)) cold_path = argv[i] + 5;
        else if (!strncmp(argv[i], "n=", 2)) cold_n = atol(argv[i] + 2);
        else if (!strncmp(argv[i], "warm=", 5)) cold_warm = atol(argv[i] + 5);
        else if (!strncmp(argv[i], "host_ms=", 8)) host_ms = atol(argv[i] + 8);
        else if (!strncmp(argv[i], "gap_us=", 7)) gaps = list_of(argv[i] + 7);
        else { fprintf(stderr, "usage: seam_bench [sizes=a,b] [threads=1,4,8] [calls=500] [reps=5]\n"); return 2; }
    }
    if (!gen_path.empty()) {                           // inputs for the cold runs, written by a process of their own
        std::vector<reef_affine> bases((size_t)cold_n);
        std::vector<reef_fe> sc((size_t)cold_n);
        CK(reef_gen_bases(REEF_PALLAS, 4242, 7, (size_t)cold_n, bases.data(), REEF_HOST));
        CK(reef_gen_scalars(REEF_PALLAS, 77, 0, 0, (size_t)cold_n, true, sc.data(), REEF_HOST));
        FILE *f = fopen(gen_path.c_str(), "wb");
        if (!f || fwrite(bases.data(), sizeof(reef_affine), bases.size(), f) != bases.size() || fwrite(sc.data(), sizeof(reef_fe), sc.size(), f) != sc.size()) { fprintf(stderr, "seam_bench: cannot write %s\n", gen_path.c_str()); return 3; }
        fclose(f);
        return 0;
    }
    if (!cold_path.empty()) {
        // A FRESH process, as `reef --prove` is: nothing of HIP has run when main() starts.  t = 0 is here.  The prover's own start-up (regex -> SAFA, the
        // step circuit: hundreds of ms and more on the host, src/backend/framework.rs:81-166) is `host_ms` of sleep; then the first commitments, one by one.
        // warm: 0 = nothing (the first call pays for the runtime), 2 = reef_runtime_init({warm = REEF_WARM_BACKGROUND}) first thing in main().
        const auto t_start = clk::now();
        reef_runtime_opts ro = {};
        ro.hw_queues = 8;
        ro.warm = (uint32_t)cold_warm;
        reef_runtime_info ri = {};
        (void)reef_runtime_init(&ro, &ri);
        const double init_ms = std::chrono::duration<double, std::milli>(clk::now() - t_start).count();
        std::vector<reef_affine> bases((size_t)cold_n);
        std::vector<reef_fe> sc((size_t)cold_n);
        FILE *f = fopen(cold_path.c_str(), "rb");
        if (!f || fread(bases.data(), sizeof(reef_affine), bases.size(), f) != bases.size() || fread(sc.data(), sizeof(reef_fe), sc.size(), f) != sc.size()) { fprintf(stderr, "seam_bench: cannot read %s (write it with gen=)\n", cold_path.c_str()); return 3; }
        fclose(f);
        std::this_thread::sleep_for(std::chrono::milliseconds(host_ms));
        double ms[6];
        reef_jacobian outs[6];
        for (int i = 0; i < 6; ++i) {
            const auto t0 = clk::now();
            mult_pippenger_pallas(&outs[i], bases.data(), (size_t)cold_n, sc.data(), true);
            ms[i] = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        (void)reef_runtime_init(nullptr, &ri);
        reef_affine aff[6];
        CK(reef_normalize(REEF_PALLAS, outs, 6, REEF_HOST, aff, nullptr));
        bool same = true;
        for (int i = 1; i < 6; ++i) same = same && memcmp(&aff[i], &aff[0], sizeof aff[0]) == 0;
        double sum = 0;
        for (double v : ms) sum += v;
        printf("{\"what\": \"a fresh process: the first six commitments\", \"n\": %ld, \"warm\": %ld, \"host_work_before_the_first_call_ms\": %ld, \"reef_runtime_init_ms\": %.3f, "
               "\"call_ms\": [%.3f, %.3f, %.3f, %.3f, %.3f, %.3f], \"sum_ms\": %.3f, \"warm_state_afterwards\": %u, \"results_identical\": %s}\n",
               cold_n, cold_warm, host_ms, init_ms, ms[0], ms[1], ms[2], ms[3], ms[4], ms[5], sum, ri.warm, same ? "true" : "false");
        return same ? 0 : 4;
    }
    {
        reef_runtime_opts ro = {};
        ro.hw_queues = 8;
        (void)reef_runtime_init(&ro, nullptr);
    }
    if (reef_device_count() < 1) { fprintf(stderr, "seam_bench: no GPU\n"); return 3; }
    if (first) {
        // What a proof that folds 1-6 times pays at the seam: the FIRST calls on a fresh key, one by one.  `gap_us` of host time
        // between the calls stands for the prover's own work between two commitments (witness synthesis, the NIFS fold; with 0 the calls are
        // back to back -- the worst case for the builder thread, which then works beside the caller).  Every call takes the same scalars, so
        // all results must be bit-identical whichever path served them (plain, or resident once the builder has published the key).
        for (long gap : gaps)
            for (long n : sizes) {
                std::vector<reef_affine> bases((size_t)n);
                CK(reef_gen_bases(REEF_PALLAS, 1000 + n % 89 + gap, 7, (size_t)n, bases.data(), REEF_HOST));
                std::vector<reef_fe> sc((size_t)n);
                CK(reef_gen_scalars(REEF_PALLAS, 33, 0, 0, (size_t)n, true, sc.data(), REEF_HOST));
                reef_key_cache_stats st0, st1;
                reef_key_cache_info(&st0);
                const int ncalls = 8;
                double ms[ncalls];
                reef_jacobian outs[ncalls];
                if (getenv("REEF_MSM_LOG") && atoi(getenv("REEF_MSM_LOG")) >= 2) fprintf(stderr, "seam_bench: n = %ld, gap %ld us\n", n, gap);
                for (int i = 0; i < ncalls; ++i) {
                    const auto t0 = clk::now();
                    mult_pippenger_pallas(&outs[i], bases.data(), (size_t)n, sc.data(), true);
                    ms[i] = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
                    if (getenv("REEF_MSM_LOG") && atoi(getenv("REEF_MSM_LOG")) >= 2)
                        fprintf(stderr, "seam_bench: call %d took %.3f ms, ended at %.3f ms\n", i + 1, ms[i],
                                (double)(std::chrono::duration_cast<std::chrono::nanoseconds>(clk::now().time_since_epoch()).count() % 100000000000ull) * 1e-6);
                    if (gap) std::this_thread::sleep_for(std::chrono::microseconds(gap));
                }
                reef_key_cache_wait();
                reef_key_cache_info(&st1);
                bool same = true;                      // as POINTS: the Jacobian representative depends on the order of the additions
                reef_affine aff[ncalls];
                CK(reef_normalize(REEF_PALLAS, outs, ncalls, REEF_HOST, aff, nullptr));
                for (int i = 1; i < ncalls; ++i) same = same && memcmp(&aff[i], &aff[0], sizeof aff[0]) == 0;
                double sum6 = 0, mx = 0;
                for (int i = 0; i < 6; ++i) { sum6 += ms[i]; mx = std::max(mx, ms[i]); }
                printf("{\"what\": \"first calls on a fresh key\", \"n\": %ld, \"gap_us\": %ld, \"call_ms\": [", n, gap);
                for (int i = 0; i < ncalls; ++i) printf("%s%.3f", i ? ", " : "", ms[i]);
                printf("], \"sum_first_6_ms\": %.3f, \"max_of_first_6_ms\": %.3f, \"builds\": %llu, \"hits\": %llu, \"spares\": %llu, \"clones\": %llu, \"results_identical\": %s}\n",
                       sum6, mx, (unsigned long long)(st1.builds - st0.builds), (unsigned long long)(st1.hits - st0.hits), (unsigned long long)(st1.spares - st0.spares),
                       (unsigned long long)(st1.clones - st0.clones), same ? "true" : "false");
                fflush(stdout);
                if (!same) return 4;
            }
        return 0;
    }
    for (long n : sizes) {
        std::vector<reef_affine> bases((size_t)n);
        CK(reef_gen_bases(REEF_PALLAS, 11 + n % 97, 3, (size_t)n, bases.data(), REEF_HOST));
        std::vector<std::vector<reef_fe>> scs(8, std::vector<reef_fe>((size_t)n));          // ordinary (pageable) host memory, as a Rust Vec is
        for (int j = 0; j < 8; ++j) CK(reef_gen_scalars(REEF_PALLAS, 20 + j, 0, 0, (size_t)n, true, scs[j].data(), REEF_HOST));
        reef_jacobian out;
        for (int i = 0; i < 4; ++i) mult_pippenger_pallas(&out, bases.data(), (size_t)n, scs[0].data(), true);   // resident from the third call on
        for (long nt : threads) {
            std::vector<double> rates, lats;
            reef_key_cache_timing tm;
            reef_key_cache_timing_get(nullptr, 1);
            for (long rep = 0; rep < reps; ++rep) {
                std::atomic<int> ready{0};
                std::atomic<bool> go{false};
                std::vector<double> lat((size_t)nt);
                std::vector<std::thread> th;
                for (long t = 0; t < nt; ++t)
                    th.emplace_back([&, t] {
                        reef_jacobian o;
                        mult_pippenger_pallas(&o, bases.data(), (size_t)n, scs[t % 8].data(), true);      // this thread's context exists
                        ready.fetch_add(1);
                        while (!go.load()) std::this_thread::yield();
                        const auto t0 = clk::now();
                        for (long c = 0; c < calls; ++c) mult_pippenger_pallas(&o, bases.data(), (size_t)n, scs[(t + c) % 8].data(), true);
                        lat[(size_t)t] = std::chrono::duration<double, std::milli>(clk::now() - t0).count() / calls;
                    });
                while (ready.load() < nt) std::this_thread::yield();
                reef_key_cache_timing_get(nullptr, rep == 0);                                   // the warm-up calls are not the measurement's
                const auto t0 = clk::now();
                go.store(true);
                for (auto &x : th) x.join();
                const double wall = std::chrono::duration<double>(clk::now() - t0).count();
                rates.push_back((double)nt * calls * n / wall / 1e6);
                double s = 0;
                for (double v : lat) s += v;
                lats.push_back(s / nt);
            }
            reef_key_cache_timing_get(&tm, 0);
            std::sort(rates.begin(), rates.end());
            std::sort(lats.begin(), lats.end());
            const double c = (double)std::max<uint64_t>(1, tm.calls);
            printf("{\"symbol\": \"mult_pippenger_pallas\", \"points\": %ld, \"threads\": %ld, \"calls_per_thread\": %ld, \"reps\": %ld, \"mpairs_per_s_median\": %.1f, \"mpairs_per_s_min\": %.1f, "
                   "\"mpairs_per_s_max\": %.1f, \"ms_per_call_median\": %.3f, \"ms_per_call_min\": %.3f, \"ms_per_call_max\": %.3f, \"host_us_per_call\": {\"nominate\": %.1f, \"enqueue\": %.1f, "
                   "\"confirm\": %.1f, \"wait\": %.1f}}\n",
                   n, nt, calls, reps, rates[rates.size() / 2], rates.front(), rates.back(), lats[lats.size() / 2], lats.front(), lats.back(), tm.nominate_ns / c / 1e3, tm.enqueue_ns / c / 1e3,
                   tm.confirm_ns / c / 1e3, tm.wait_ns / c / 1e3);
            fflush(stdout);
        }
    }
    reef_key_cache_stats st;
    reef_key_cache_info(&st);
    printf("{\"key_table\": {\"resident_keys\": %llu, \"resident_mib\": %.0f, \"builds\": %llu, \"clones\": %llu, \"hits\": %llu, \"misspeculated\": %llu}}\n", (unsigned long long)st.resident_keys,
           st.resident_bytes / 1048576.0, (unsigned long long)st.builds, (unsigned long long)st.clones, (unsigned long long)st.hits, (unsigned long long)st.misspeculated);
    return 0;
}
