// reef_replay -- issues, through the C ABI only, the MSM sequence of one `reef --prove` run
// (eniac/Reef src/backend/framework.rs:642-754) with synthetic scalars of the right shapes, and
// times it.  Reef itself is Rust and cannot be built in this image (no cargo, crates not
// vendored), so the "--prove" numbers of this backend are REPLAYS of the commitment work, never
// an end-to-end proof: NFA construction, witness generation and sum-checks stay on the host and
// are not part of what is timed here.
//
// Sequence (sizes from Reef's cost model, src/backend/costs.rs):
//   setup            keys for G1 (Pallas) and G2 (Vesta): generated + pre-shifted on the GPU
//   per folding step comm_T2 (|C2| Vesta) -> comm_W1 (|W1| Pallas) -> comm_T1 (|C1| Pallas)
//                    -> comm_W2 (|W2| Vesta); each commitment feeds the next circuit's public
//                    input, so the four MSMs are issued one after the other, result to host
//   final SNARK      one more |C2| MSM, then for each curve an IPA over the padded key length:
//                    log2 N rounds of { L, R cross MSMs (issued on two streams), generator fold }
//   consistency      IPA of the Hyrax row length (prove_eval, commitment.rs:371/383)
//   devices=N        the multi-device leg (replay_devices): arguments placed whole on N members, the document commitment through a device group
// and, beside the MSMs: the Hyrax commitment of the document (--commit), one nlookup sum-check per
// folding step (rows N2) and the document polynomial's row binding at proof end (row N3).
//
// What crosses PCIe: the per-step scalar vectors live in ordinary HOST memory, as nova hands them over
// (framework.rs:668), and every commitment comes back to the host.  What is checked: the generators are an
// arithmetic progression B_i = (k0 + i*d)*G, so each MSM has a known discrete logarithm; every per-step
// commitment is compared with (sum_i s_i*(k0 + i*d) mod r)*G computed from host big-integer arithmetic and a
// one-point key, and the replay aborts on the first mismatch.  The sizes are PREDICTIONS of Reef's cost model
// (src/backend/costs.rs), not measurements of a Reef run: flagged in the JSON line.  No MSM length is typed in here: the
// shapes are read from tests/golden/replay_shapes.json, which oracle/gen_replay_shapes.py derives from its restatement of
// costs.rs (the SAFA shape of each regex is an input of that script).
//
// One translation unit, two artefacts (csrc/Makefile): libreef_replay.so exports reef_replay_run() -- bench.py calls it
// in-process after its timed region, tests/test_gpu_replay.py under pytest -- and the reef_replay executable is its main().
// Build: g++ -O2 -std=c++17 -fPIC -shared reef_replay.cpp -I../../../include -L../../_lib -lreef_msm -o libreef_replay.so
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "reef_msm.h"
#include "replay_standins.h"

// a failed call ends the replay with its message (reef_replay_run returns it; the executable prints it and exits non-zero)
[[noreturn]] static void fail(const std::string &msg) { throw std::runtime_error(msg); }
#define CK(x)                                                                              \
    do {                                                                                   \
        reef_status s_ = (x);                                                              \
        if (s_ != REEF_OK) fail(std::string(#x) + " failed: " + reef_last_error());        \
    } while (0)

// Everything the replay allocates is owned by one of these: a failed call (an exception) unwinds through them, so a replay that
// ends early inside bench.py's process leaves no key and no device buffer behind.
struct DevFree { void operator()(void *p) const { if (p) reef_device_free(p); } };
template <class T> using dev_ptr = std::unique_ptr<T, DevFree>;
struct CtxFree { void operator()(reef_msm_ctx *p) const { if (p) reef_msm_ctx_destroy(p); } };
using ctx_ptr = std::unique_ptr<reef_msm_ctx, CtxFree>;
struct ScFree { void operator()(reef_sc_ctx *p) const { if (p) reef_sc_destroy(p); } };
using sc_ptr = std::unique_ptr<reef_sc_ctx, ScFree>;
template <class T> static dev_ptr<T> device_alloc(size_t count) {
    T *p = (T *)reef_device_alloc(count * sizeof(T));
    if (!p) throw std::runtime_error(std::string("alloc: ") + reef_last_error());
    return dev_ptr<T>(p);
}

using clk = std::chrono::steady_clock;
static double ms_since(clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); }

// ---- host-side check: discrete logarithm of an MSM over generators in arithmetic progression ------------------
static const uint64_t ORDER[2][4] = {   // group orders: Pallas (= Fq), Vesta (= Fp); little-endian limbs
    {0x8c46eb2100000001ULL, 0x224698fc0994a8ddULL, 0x0ULL, 0x4000000000000000ULL},
    {0x992d30ed00000001ULL, 0x224698fc094cf91bULL, 0x0ULL, 0x4000000000000000ULL}};
struct Big { uint64_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0}; };
static void big_addmul(Big &a, const uint64_t s[4], uint64_t m) {   // a += s * m
    unsigned __int128 carry = 0;
    for (int i = 0; i < 8; ++i) {
        unsigned __int128 t = (unsigned __int128)a.w[i] + carry + (i < 4 ? (unsigned __int128)s[i] * m : 0);
        a.w[i] = (uint64_t)t;
        carry = t >> 64;
    }
}
static int big_cmp_shifted(const Big &a, const uint64_t r[4], int shift) {   // a <=> r << shift
    Big b;
    const int ws = shift / 64, bs = shift % 64;
    for (int i = 0; i < 4; ++i) {
        if (i + ws < 8) b.w[i + ws] |= r[i] << bs;
        if (bs && i + ws + 1 < 8) b.w[i + ws + 1] |= r[i] >> (64 - bs);
    }
    for (int i = 7; i >= 0; --i)
        if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
    return 0;
}
static void big_sub_shifted(Big &a, const uint64_t r[4], int shift) {
    Big b;
    const int ws = shift / 64, bs = shift % 64;
    for (int i = 0; i < 4; ++i) {
        if (i + ws < 8) b.w[i + ws] |= r[i] << bs;
        if (bs && i + ws + 1 < 8) b.w[i + ws + 1] |= r[i] >> (64 - bs);
    }
    unsigned __int128 borrow = 0;
    for (int i = 0; i < 8; ++i) {
        unsigned __int128 t = (unsigned __int128)a.w[i] - b.w[i] - borrow;
        a.w[i] = (uint64_t)t;
        borrow = (t >> 64) & 1;
    }
}
// (sum_i canon[i] * (k0 + i*d)) mod r as a canonical 4-limb scalar
static reef_fe dlog_of_msm(int curve, const reef_fe *canon, size_t n, uint64_t k0, uint64_t d) {
    Big acc;
    for (size_t i = 0; i < n; ++i) big_addmul(acc, canon[i].l, k0 + (uint64_t)i * d);
    for (int shift = 511 - 255; shift >= 0; --shift)
        if (big_cmp_shifted(acc, ORDER[curve], shift) >= 0) big_sub_shifted(acc, ORDER[curve], shift);
    reef_fe r;
    for (int i = 0; i < 4; ++i) r.l[i] = acc.w[i];
    return r;
}

struct Shape {
    std::string name;
    size_t w1 = 0, c1 = 0, w2 = 0, c2 = 0;   // |W1|, |C1| (Pallas), |W2|, |C2| (Vesta)
    int steps = 0;
    size_t hyrax_row = 0;    // R = 2^(l - l/2): length of the consistency IPA (0 = merkle mode)
    int doc_log = 0;         // l = log2 of the padded document length the Hyrax commitment covers (0: no Hyrax commitment)
    int symbol_bits = 0;     // width of a document symbol (alphabet + EOF/EPSILON, framework.rs:978-1011)
    int table_log = 0;       // log2 of the table the per-step nlookup sum-check runs over (r1cs.rs:2318-2385); 0: not replayed
    int lookups = 0;         // lookups folded per step (batch size)
    int merkle_log = 0;      // --merkle: log2 of the document the Poseidon tree commits to (0: Hyrax commitment)
};

// ---- tests/golden/replay_shapes.json: the "shapes" array of flat objects written by oracle/gen_replay_shapes.py ----
static std::string read_file(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) fail("cannot open the replay shapes file " + path + " (generated by oracle/gen_replay_shapes.py)");
    std::string s;
    char buf[4096];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, got);
    fclose(f);
    return s;
}
static long long json_int(const std::string &obj, const char *key) {
    const std::string k = std::string("\"") + key + "\":";
    size_t p = obj.find(k);
    if (p == std::string::npos) fail(std::string("replay shapes: field ") + key + " missing");
    return atoll(obj.c_str() + p + k.size());
}
static Shape load_shape(const std::string &path, const char *which) {
    const std::string text = read_file(path);
    size_t p = text.find("\"shapes\":");
    if (p == std::string::npos) fail("replay shapes: no \"shapes\" array in " + path);
    for (;;) {
        const size_t b = text.find('{', p);
        if (b == std::string::npos) break;
        const size_t e = text.find('}', b);
        if (e == std::string::npos) break;
        const std::string obj = text.substr(b, e - b + 1);
        p = e + 1;
        const size_t np = obj.find("\"name\":");
        if (np == std::string::npos) continue;
        const size_t q0 = obj.find('"', np + 7), q1 = obj.find('"', q0 + 1);
        const std::string name = obj.substr(q0 + 1, q1 - q0 - 1);
        if (name.find(which) == std::string::npos) continue;
        Shape s;
        s.name = name;
        s.w1 = (size_t)json_int(obj, "w1"); s.c1 = (size_t)json_int(obj, "c1");
        s.w2 = (size_t)json_int(obj, "w2"); s.c2 = (size_t)json_int(obj, "c2");
        s.steps = (int)json_int(obj, "steps");
        s.hyrax_row = (size_t)json_int(obj, "hyrax_row");
        s.doc_log = (int)json_int(obj, "doc_log");
        s.symbol_bits = (int)json_int(obj, "symbol_bits");
        s.table_log = (int)json_int(obj, "table_log");
        s.lookups = (int)json_int(obj, "lookups");
        s.merkle_log = (int)json_int(obj, "merkle_log");
        if (!s.w1 || !s.c1 || !s.w2 || !s.c2 || s.steps < 1) fail("replay shapes: " + name + " is not a usable shape");
        return s;
    }
    fail(std::string("replay shapes: no config matching '") + which + "' in " + path);
}

struct Curve {
    int id;
    reef_msm_ctx *key = nullptr;      // pre-shifted resident commitment key
    reef_msm_ctx *ipa[2] = {nullptr, nullptr};  // plain contexts re-keyed every IPA round
    reef_affine *d_gens = nullptr;    // device copy of the key for the IPA
    size_t n = 0;
    uint64_t k0 = 0, d = 0;           // generators B_i = (k0 + i*d)*G
    reef_msm_ctx *one = nullptr;      // the one-point key [G]: turns an expected discrete logarithm into a point
};
static int g_checked = 0;
// abort unless `got` is dlog*G (both normalised to affine)
static void check_point(Curve &c, const reef_jacobian &got, const reef_fe &dlog, const char *what) {
    reef_jacobian want;
    CK(reef_msm(c.one, &dlog, 1, REEF_HOST, false, &want, REEF_HOST));
    reef_jacobian both[2] = {got, want};
    reef_affine aff[2];
    CK(reef_normalize(c.id, both, 2, REEF_HOST, aff, nullptr));
    if (memcmp(&aff[0], &aff[1], sizeof(reef_affine)) != 0)
        fail(std::string("reef_replay: ") + what + " on curve " + std::to_string(c.id) + " differs from its discrete-logarithm closed form");
    ++g_checked;
}

static size_t next_pow2(size_t x) { size_t p = 1; while (p < x) p <<= 1; return p; }

static dev_ptr<reef_fe> device_scalars(int curve, uint64_t seed, int kind, size_t n) {
    dev_ptr<reef_fe> p = device_alloc<reef_fe>(n);
    CK(reef_gen_scalars(curve, seed, kind, 0, n, true, p.get(), REEF_DEVICE));
    return p;
}

// One IPA without generator folding: every round's cross terms are two MSMs over the ORIGINAL
// pre-shifted key with scalars a[.] * prod(challenges) (reef_ipa_cross_terms); the vector fold
// a' = r*a_lo + r^-1*a_hi is field-only host work in nova and is not replayed.
static double run_ipa_nofold(Curve &c, size_t n, const reef_fe *d_scalars, int *rounds_out) {
    auto t0 = clk::now();
    std::vector<reef_fe> w1s, w2s;
    reef_jacobian L, R;
    int rounds = 0;
    for (size_t len = n; len > 1; len /= 2, ++rounds) {
        CK(reef_ipa_cross_terms(c.key, d_scalars, len, REEF_DEVICE, true, w1s.data(), w2s.data(), w1s.size(), &L, &R));
        reef_fe w1 = {{0x1234567890abcdefULL + rounds, 0x0fedcba987654321ULL, 0x1111111122222222ULL, 0x0333333344444444ULL}};
        reef_fe w2 = {{0x0badc0ffee0ddf00ULL + rounds, 0x0123456789abcdefULL, 0x5555555566666666ULL, 0x0777777788888888ULL}};
        w1s.push_back(w1);
        w2s.push_back(w2);
    }
    if (rounds_out) *rounds_out = rounds;
    return ms_since(t0);
}

// One IPA: log2(n) rounds, two cross MSMs of n/2 points on two streams + the generator fold.
static double run_ipa(Curve &c, size_t n, const reef_fe *d_scalars, int *rounds_out) {
    auto t0 = clk::now();
    reef_affine *cur = c.d_gens;
    dev_ptr<reef_affine> buf[2] = {device_alloc<reef_affine>(n / 2 + 1), device_alloc<reef_affine>(n / 2 + 1)};
    reef_jacobian L, R;
    reef_fe w1 = {{0x1234567890abcdefULL, 0x0fedcba987654321ULL, 0x1111111122222222ULL, 0x0333333344444444ULL}};
    reef_fe w2 = {{0x0badc0ffee0ddf00ULL, 0x0123456789abcdefULL, 0x5555555566666666ULL, 0x0777777788888888ULL}};
    int rounds = 0, flip = 0;
    for (size_t len = n; len > 1; len /= 2, ++rounds) {
        const size_t half = len / 2;
        CK(reef_msm_ctx_set_bases(c.ipa[0], cur + half, half, REEF_DEVICE));   // L = <a_lo, G_hi>
        CK(reef_msm_ctx_set_bases(c.ipa[1], cur, half, REEF_DEVICE));          // R = <a_hi, G_lo>
        // the cross terms feed the transcript, so they go straight to the host: for keys without
        // pre-shifted tables the library then finishes the window combine on a host core
        CK(reef_msm(c.ipa[0], d_scalars, half, REEF_DEVICE, true, &L, REEF_HOST));
        CK(reef_msm(c.ipa[1], d_scalars + half, half, REEF_DEVICE, true, &R, REEF_HOST));
        CK(reef_fold(c.id, cur, half, REEF_DEVICE, &w1, &w2, buf[flip].get()));   // G' = w1*G_lo + w2*G_hi
        cur = buf[flip].get();
        flip ^= 1;
    }
    if (rounds_out) *rounds_out = rounds;
    return ms_since(t0);
}

// ---- the work around the MSMs that this backend also covers -------------------------------------
// --commit: HyraxPC::commit over the document matrix (commitment.rs:187), from the document bytes.
// per step: the nlookup sum-check of witness generation (r1cs.rs:2318-2385) as reef_sc_* rounds, the
//           Poseidon challenge of every round replaced by a fixed field element (it stays on the host).
// proof end: doc_poly.evaluate / the row binding of prove_eval (commitment.rs:357,371-391).
static dev_ptr<uint8_t> device_symbols(size_t n, int bits, uint64_t seed) {
    std::vector<uint8_t> h(n);
    uint64_t x = seed;
    const uint32_t bound = bits >= 8 ? 131u : (bits == 3 ? 7u : (1u << bits));
    for (size_t i = 0; i < n; ++i) {
        x = x * 6364136223846793005ULL + 1442695040888963407ULL;
        h[i] = (uint8_t)((x >> 33) % bound);
    }
    dev_ptr<uint8_t> d = device_alloc<uint8_t>(n);
    CK(reef_memcpy(d.get(), h.data(), n, REEF_DEVICE, REEF_HOST));
    return d;
}

static double run_hyrax_commit(const Shape *sh, reef_affine *d_gens, const uint8_t *d_doc, double *first_ms) {
    const size_t rows = (size_t)1 << (sh->doc_log / 2), row_len = (size_t)1 << (sh->doc_log - sh->doc_log / 2);
    reef_msm_ctx *key = nullptr;
    CK(reef_msm_ctx_create(&key, REEF_PALLAS, d_gens, row_len, REEF_DEVICE, nullptr));
    const ctx_ptr key_owner(key);
    const dev_ptr<reef_jacobian> d_out_owner = device_alloc<reef_jacobian>(rows);
    reef_jacobian *d_out = d_out_owner.get();
    auto t0 = clk::now();
    CK(reef_msm_rows_symbols(key, d_doc, rows, row_len, REEF_DEVICE, (uint32_t)sh->symbol_bits, nullptr, nullptr, true, d_out, REEF_DEVICE));
    CK(reef_msm_ctx_sync(key));
    *first_ms = ms_since(t0);                          // includes the construction of the symbol tables
    t0 = clk::now();
    CK(reef_msm_rows_symbols(key, d_doc, rows, row_len, REEF_DEVICE, (uint32_t)sh->symbol_bits, nullptr, nullptr, true, d_out, REEF_DEVICE));
    CK(reef_msm_ctx_sync(key));
    const double again = ms_since(t0);
    return again;
}

static double run_sumcheck_step(reef_sc_ctx *sc, int ell, int lookups) {
    std::vector<reef_fe> rs(lookups + 1), last_q(ell);
    std::vector<uint32_t> qs(lookups);
    for (int i = 0; i <= lookups; ++i) rs[i] = reef_fe{{0x9e3779b97f4a7c15ULL * (i + 1), 0x1234ULL + i, 0, 0}};
    for (int i = 0; i < lookups; ++i) qs[i] = (uint32_t)((0x2545F4914F6CDD1DULL * (i + 7)) >> (64 - ell));
    for (int j = 0; j < ell; ++j) last_q[j] = reef_fe{{0xabcdef12345ULL + j, 0x77ULL * j, 0, 0}};
    auto t0 = clk::now();
    CK(reef_sc_reset_table(sc));
    CK(reef_sc_gen_eq_table(sc, rs.data(), qs.data(), lookups, last_q.data(), ell));
    reef_fe g[3];
    CK(reef_sc_round_coeffs(sc, (size_t)1 << (ell - 1), g));
    for (int i = 1; i <= ell; ++i) {
        const reef_fe r = {{0x5851f42d4c957f2dULL + i, 0x14057b7ef767814fULL, 0x0123456789abcdefULL, 0x0fedcba987654321ULL}};
        const size_t pow = (size_t)1 << (ell - i);
        if (pow >= 2) CK(reef_sc_fold_and_next_coeffs(sc, pow, &r, g));
        else CK(reef_sc_fold(sc, pow, &r));
    }
    reef_fe v;
    CK(reef_sc_read(sc, 0, 1, &v));                    // next_running_v (r1cs.rs:2379-2385)
    return ms_since(t0);
}


// ---- the multi-device leg: what ONE prover process can hand to the other GPUs of its node (include/reef_msm.h section 5) ----
// Reef's per-step MSMs are 2^14-2^16 points and are not split.  What a node can take from a --prove run is WHOLE units:
//   * the final SNARK's three arguments (two Spartan IPAs, the consistency IPA; src/backend/framework.rs:695-721) on up to three
//     devices, each argument's key and scalars resident on its device (reef_msm_opts.device), issued from three caller threads;
//   * the Hyrax commitment of the document (src/backend/commitment.rs:187) through a device group (reef_msm_group_rows_symbols:
//     rows dealt out whole, the document in host memory as Reef holds it), checked row for row against one device.
// `ordinals` may repeat a device (a one-GPU box runs the whole leg on device 0; the JSON says how many devices were distinct).
struct DeviceScope {
    int prev = 0;
    explicit DeviceScope(int dev) {
        CK(reef_get_device(&prev));
        CK(reef_set_device(dev));
    }
    ~DeviceScope() { (void)reef_set_device(prev); }
};
struct GroupFree { void operator()(reef_msm_group *p) const { if (p) reef_msm_group_destroy(p); } };
static std::string replay_devices(const Shape &shape, const std::vector<int> &ordinals, bool tables) {
    const Shape *sh = &shape;
    const size_t nd = ordinals.size();
    std::vector<int> uniq(ordinals);
    std::sort(uniq.begin(), uniq.end());
    uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
    struct Arg { const char *name; Curve c; int device; dev_ptr<reef_affine> gens; dev_ptr<reef_fe> sc; ctx_ptr key; double alone_ms = 0; };
    std::vector<Arg> args(3);
    args[0].name = "ipa_pallas"; args[0].c.id = REEF_PALLAS; args[0].c.n = next_pow2(std::max(sh->w1, sh->c1));
    args[1].name = "ipa_vesta";  args[1].c.id = REEF_VESTA;  args[1].c.n = next_pow2(std::max(sh->w2, sh->c2));
    args[2].name = "consistency"; args[2].c.id = REEF_PALLAS; args[2].c.n = sh->hyrax_row;
    if (sh->hyrax_row < 2) args.pop_back();
    // longest argument first onto the least loaded device (deterministic; the same rule as reef_amd/distributed.py::place_units)
    std::vector<size_t> order(args.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return args[a].c.n > args[b].c.n; });
    std::vector<double> load(nd, 0.0);
    for (size_t k : order) {
        size_t best = 0;
        for (size_t d = 1; d < nd; ++d)
            if (load[d] < load[best]) best = d;
        args[k].device = ordinals[best];
        load[best] += (double)args[k].c.n;
    }
    for (Arg &a : args) {
        DeviceScope ds(a.device);
        a.gens = device_alloc<reef_affine>(a.c.n);
        CK(reef_gen_bases(a.c.id, 0xC0FFEE + a.c.id, 7, a.c.n, a.gens.get(), REEF_DEVICE));
        a.sc = device_scalars(a.c.id, 12 + a.c.id, 0, a.c.n);
        reef_msm_opts o = {};
        o.bucket_groups = 1;
        o.byte_tables = tables ? 1 : 2;
        o.device = a.device;
        CK(reef_msm_ctx_create(&a.c.key, a.c.id, a.gens.get(), a.c.n, REEF_DEVICE, &o));
        a.key.reset(a.c.key);
        run_ipa_nofold(a.c, a.c.n, a.sc.get(), nullptr);                     // warm-up: workspaces
        a.alone_ms = run_ipa_nofold(a.c, a.c.n, a.sc.get(), nullptr);
    }
    double together_ms = 0;
    {
        std::vector<std::exception_ptr> err(args.size());
        std::vector<std::thread> th;
        auto tc = clk::now();
        for (size_t k = 1; k < args.size(); ++k)
            th.emplace_back([&, k] { try { run_ipa_nofold(args[k].c, args[k].c.n, args[k].sc.get(), nullptr); } catch (...) { err[k] = std::current_exception(); } });
        try { run_ipa_nofold(args[0].c, args[0].c.n, args[0].sc.get(), nullptr); } catch (...) { err[0] = std::current_exception(); }
        for (auto &t : th) t.join();
        together_ms = ms_since(tc);
        for (auto &e : err)
            if (e) std::rethrow_exception(e);
    }
    // the document commitment over the group, against one device
    double commit_group_ms = 0, commit_one_ms = 0;
    reef_msm_group_info gi;
    memset(&gi, 0, sizeof gi);
    if (sh->doc_log) {
        const size_t n_doc = (size_t)1 << sh->doc_log;
        const size_t rows = (size_t)1 << (sh->doc_log / 2), row_len = (size_t)1 << (sh->doc_log - sh->doc_log / 2);
        std::vector<uint8_t> doc(n_doc);
        uint64_t x = 0xD0C;
        const uint32_t bound = sh->symbol_bits >= 8 ? 131u : (sh->symbol_bits == 3 ? 7u : (1u << sh->symbol_bits));
        for (size_t i = 0; i < n_doc; ++i) { x = x * 6364136223846793005ULL + 1442695040888963407ULL; doc[i] = (uint8_t)((x >> 33) % bound); }
        std::vector<reef_affine> gens(row_len);
        CK(reef_gen_bases(REEF_PALLAS, 0xFEED, 3, row_len, gens.data(), REEF_HOST));
        reef_msm_group *grp = nullptr;
        reef_msm_group_opts go = {};
        go.split = REEF_SPLIT_WINDOWS;
        CK(reef_msm_group_create(&grp, REEF_PALLAS, gens.data(), row_len, REEF_HOST, nullptr, ordinals.data(), nd, &go));
        const std::unique_ptr<reef_msm_group, GroupFree> grp_owner(grp);
        CK(reef_msm_group_info_get(grp, &gi));
        std::vector<reef_jacobian> out_g(rows), out_1(rows);
        CK(reef_msm_group_rows_symbols(grp, doc.data(), rows, row_len, REEF_HOST, (uint32_t)sh->symbol_bits, nullptr, nullptr, true, out_g.data()));   // builds the symbol tables
        // the median of five: the first pageable copy a pool stream carries pays for the runtime's staging buffers (one-off, 5-20 ms),
        // and the members take whichever stream is least busy
        auto median5 = [&](const std::function<void()> &f) {
            double t[5];
            for (double &x : t) { auto t0 = clk::now(); f(); x = ms_since(t0); }
            std::sort(t, t + 5);
            return t[2];
        };
        commit_group_ms = median5([&] { CK(reef_msm_group_rows_symbols(grp, doc.data(), rows, row_len, REEF_HOST, (uint32_t)sh->symbol_bits, nullptr, nullptr, true, out_g.data())); });
        reef_msm_ctx *one = nullptr;
        reef_msm_opts o1 = {};
        o1.device = ordinals[0];
        CK(reef_msm_ctx_create(&one, REEF_PALLAS, gens.data(), row_len, REEF_HOST, &o1));
        const ctx_ptr one_owner(one);
        CK(reef_msm_rows_symbols(one, doc.data(), rows, row_len, REEF_HOST, (uint32_t)sh->symbol_bits, nullptr, nullptr, true, out_1.data(), REEF_HOST));
        commit_one_ms = median5([&] { CK(reef_msm_rows_symbols(one, doc.data(), rows, row_len, REEF_HOST, (uint32_t)sh->symbol_bits, nullptr, nullptr, true, out_1.data(), REEF_HOST)); });
        std::vector<reef_affine> ag(rows), a1(rows);
        CK(reef_normalize(REEF_PALLAS, out_g.data(), rows, REEF_HOST, ag.data(), nullptr));
        CK(reef_normalize(REEF_PALLAS, out_1.data(), rows, REEF_HOST, a1.data(), nullptr));
        if (memcmp(ag.data(), a1.data(), rows * sizeof(reef_affine)) != 0) fail("reef_replay: the group's row commitments differ from one device's");
    }
    // --merkle: the Poseidon tree of the document in blocks over the devices (reef_merkle_commit_devices), root against one device's
    double merkle_devices_ms = 0, merkle_one_ms = 0;
    uint32_t merkle_blocks = 0;
    if (sh->merkle_log) {
        const size_t n_doc = (size_t)1 << sh->merkle_log;
        std::vector<uint32_t> doc(n_doc);
        for (size_t i = 0; i < n_doc; ++i) doc[i] = (uint32_t)((i * 2654435761u) >> 24);
        reef_poseidon_params pp;
        pp.width = 5; pp.full_rounds = STANDIN_POSEIDON_RF; pp.partial_rounds = STANDIN_POSEIDON_RP; pp.reserved = 0;
        pp.round_constants = STANDIN_POSEIDON_RC; pp.mds = STANDIN_POSEIDON_MDS;
        pp.tag_leaf = STANDIN_POSEIDON_TAGS[0]; pp.tag_node = STANDIN_POSEIDON_TAGS[1];
        reef_fe root_d, root_1;
        CK(reef_merkle_commit_devices(REEF_PALLAS, &pp, doc.data(), std::min(n_doc, (size_t)1 << 16), false, ordinals.data(), nd, nullptr, &root_d, nullptr));      // warm-up (never beyond the document)
        auto t0 = clk::now();
        CK(reef_merkle_commit_devices(REEF_PALLAS, &pp, doc.data(), n_doc, false, ordinals.data(), nd, nullptr, &root_d, &merkle_blocks));
        merkle_devices_ms = ms_since(t0);
        t0 = clk::now();
        CK(reef_merkle_commit(REEF_PALLAS, &pp, doc.data(), n_doc, REEF_HOST, false, nullptr, REEF_HOST, &root_1));
        merkle_one_ms = ms_since(t0);
        if (memcmp(&root_d, &root_1, sizeof root_1) != 0) fail("reef_replay: the Merkle commitment built in blocks differs from one device's");
    }
    double sum_alone = 0, longest = 0;
    for (Arg &a : args) { sum_alone += a.alone_ms; longest = std::max(longest, a.alone_ms); }
    std::string placed = "[";
    for (size_t k = 0; k < args.size(); ++k) {
        char b[160];
        snprintf(b, sizeof b, "%s{\"argument\": \"%s\", \"points\": %zu, \"device\": %d, \"alone_ms\": %.3f}", k ? ", " : "", args[k].name, args[k].c.n, args[k].device, args[k].alone_ms);
        placed += b;
    }
    placed += "]";
    std::vector<char> line(4096);
    snprintf(line.data(), line.size(), "{\"members\": %zu, \"distinct_devices\": %zu, \"visible_devices\": %d, \"note\": \"one process: the final SNARK's arguments placed whole on the devices "
             "(per-device contexts, one caller thread each), the document commitment through a device group; ordinals repeat when the box has fewer GPUs than members\", "
             "\"final_snark_placed\": %s, \"three_arguments_one_after_the_other_ms\": %.3f, \"three_arguments_on_devices_ms\": %.3f, \"longest_argument_ms\": %.3f, "
             "\"commit_hyrax_group_ms\": %.3f, \"commit_hyrax_one_device_ms\": %.3f, \"commit_rows_checked_against_one_device\": %s, \"group_exchange\": \"%s\", \"group_peer_members\": %u, "
             "\"commit_merkle_devices_ms\": %.3f, \"commit_merkle_one_device_ms\": %.3f, \"commit_merkle_blocks\": %u, \"commit_merkle_root_checked_against_one_device\": %s}",
             nd, uniq.size(), reef_device_count(), placed.c_str(), sum_alone, together_ms, longest, commit_group_ms, commit_one_ms, sh->doc_log ? "true" : "false",
             gi.exchange == REEF_EXCHANGE_HOST ? "host-staged" : "peer copies (hipMemcpyPeerAsync; in place on a shared device)", gi.peer_members,
             merkle_devices_ms, merkle_one_ms, merkle_blocks, sh->merkle_log ? "true" : "false");
    return std::string(line.data());
}

// The whole replay of one config; returns the JSON line.
static std::string replay_body(const Shape &shape, bool nofold, bool tables, const std::string &shapes_path, const std::vector<int> &ordinals) {
    const Shape *sh = &shape;
    Curve cv[2];
    cv[0].id = REEF_PALLAS; cv[0].n = next_pow2(sh->w1 > sh->c1 ? sh->w1 : sh->c1);
    cv[1].id = REEF_VESTA;  cv[1].n = next_pow2(sh->w2 > sh->c2 ? sh->w2 : sh->c2);

    std::vector<dev_ptr<reef_affine>> owned_gens;       // destroyed after the contexts below
    std::vector<ctx_ptr> owned_ctx;

    // ---- set-up as a --prove run pays it.  Reef derives its commitment keys from their labels on EVERY run (src/backend/framework.rs:115,
    // 297-303 -> CommitmentGens::new), so set-up is prove time: per curve, on a caller thread of its own -- the two curves at
    // once -- label -> n generators on the device (row N1: reef_derive_generators; stand-in hash-to-curve parameters, replay_standins.h) ->
    // resident pre-shifted key straight from the device buffer (reef_msm_ctx_create): what CommitmentGens(label, n, params) of
    // reef_provider.hpp does.  Timed twice: setup_first_ms is the first pass in this process (first launches of the kernels, the
    // workspaces' allocations), setup_ms the second.  The keys built here are dropped again: the MSMs below run on keys of the same
    // sizes whose generators are an arithmetic progression (so that every commitment has a known discrete logarithm), built after the
    // clock has stopped (check_keys_ms).
    auto derive_both = [&]() {
        std::exception_ptr err[2];
        auto one = [&](int k) {
            try {
                const int id = cv[k].id;
                reef_keygen_params kp;
                const reef_fe *sp = id == REEF_PALLAS ? STANDIN_KEYGEN_0 : STANDIN_KEYGEN_1;
                const char *dst = id == REEF_PALLAS ? STANDIN_KEYGEN_DST_0 : STANDIN_KEYGEN_DST_1;
                kp.a = sp[0]; kp.b = sp[1]; kp.z = sp[2];
                memcpy(kp.iso, sp + 3, 13 * sizeof(reef_fe));
                kp.dst = (const uint8_t *)dst; kp.dst_len = (uint32_t)strlen(dst); kp.little_endian = 0;
                const dev_ptr<reef_affine> gens = device_alloc<reef_affine>(cv[k].n);
                CK(reef_derive_generators(id, (const uint8_t *)"ck", 2, cv[k].n, &kp, false, gens.get(), REEF_DEVICE));
                reef_msm_opts o = {};
                o.bucket_groups = 1;              // commitment keys are fixed for the life of PublicParams: pre-shift once
                o.byte_tables = tables ? 1 : 2;   // explicit: built with the key, or never (no switch of paths mid-run)
                o.device = -1;
                reef_msm_ctx *key = nullptr;
                CK(reef_msm_ctx_create(&key, id, gens.get(), cv[k].n, REEF_DEVICE, &o));
                const ctx_ptr owner(key);
                CK(reef_msm_ctx_sync(key));
            } catch (...) { err[k] = std::current_exception(); }
        };
        auto t0 = clk::now();
        std::thread other(one, 1);
        one(0);
        other.join();
        const double ms = ms_since(t0);
        for (auto &e : err)
            if (e) std::rethrow_exception(e);
        return ms;
    };
    const double setup_first_ms = derive_both();
    const double setup_ms = derive_both();

    auto t_check_keys = clk::now();
    for (Curve &c : cv) {
        owned_gens.push_back(device_alloc<reef_affine>(c.n));
        c.d_gens = owned_gens.back().get();
        c.k0 = 0xC0FFEE + c.id; c.d = 7;
        CK(reef_gen_bases(c.id, c.k0, c.d, c.n, c.d_gens, REEF_DEVICE));
        {
            reef_affine g1;
            CK(reef_gen_bases(c.id, 1, 0, 1, &g1, REEF_HOST));           // 1*G
            reef_msm_opts og = {};
            og.bucket_groups = 1;
            og.device = -1;
            CK(reef_msm_ctx_create(&c.one, c.id, &g1, 1, REEF_HOST, &og));
            owned_ctx.emplace_back(c.one);
        }
        reef_msm_opts o = {};
        o.bucket_groups = 1;
        o.byte_tables = tables ? 1 : 2;
        o.device = -1;
        CK(reef_msm_ctx_create(&c.key, c.id, c.d_gens, c.n, REEF_DEVICE, &o));
        owned_ctx.emplace_back(c.key);
        CK(reef_msm_ctx_sync(c.key));
        reef_msm_opts plain = {};
        plain.device = -1;
        if (!nofold)   // the per-round re-keyed contexts exist only in the generator-fold IPA
            for (auto &x : c.ipa) { CK(reef_msm_ctx_create(&x, c.id, c.d_gens, c.n / 2, REEF_DEVICE, &plain)); owned_ctx.emplace_back(x); }
    }
    const double check_keys_ms = ms_since(t_check_keys);

    // witness-like scalars for W, uniform for the cross terms T
    const dev_ptr<reef_fe> oW1 = device_scalars(REEF_PALLAS, 11, 1, cv[0].n), oT1 = device_scalars(REEF_PALLAS, 12, 0, cv[0].n);
    const dev_ptr<reef_fe> oW2 = device_scalars(REEF_VESTA, 13, 1, cv[1].n), oT2 = device_scalars(REEF_VESTA, 14, 0, cv[1].n);
    reef_fe *sW1 = oW1.get(), *sT1 = oT1.get(), *sW2 = oW2.get(), *sT2 = oT2.get();
    // the same vectors in HOST memory (Montgomery form, what nova holds) and as canonical integers (for the check)
    struct HostVec { std::vector<reef_fe> mont, canon; };
    auto host_vec = [&](int curve, uint64_t seed, int kind, size_t n) {
        HostVec v;
        v.mont.resize(n); v.canon.resize(n);
        CK(reef_gen_scalars(curve, seed, kind, 0, n, true, v.mont.data(), REEF_HOST));
        CK(reef_gen_scalars(curve, seed, kind, 0, n, false, v.canon.data(), REEF_HOST));
        return v;
    };
    HostVec hW1 = host_vec(REEF_PALLAS, 11, 1, cv[0].n), hT1 = host_vec(REEF_PALLAS, 12, 0, cv[0].n);
    HostVec hW2 = host_vec(REEF_VESTA, 13, 1, cv[1].n), hT2 = host_vec(REEF_VESTA, 14, 0, cv[1].n);
    // expected discrete logarithms of the four per-step commitments (computed once, outside the timed loops)
    const reef_fe eW1 = dlog_of_msm(0, hW1.canon.data(), sh->w1, cv[0].k0, cv[0].d), eT1 = dlog_of_msm(0, hT1.canon.data(), sh->c1, cv[0].k0, cv[0].d);
    const reef_fe eW2 = dlog_of_msm(1, hW2.canon.data(), sh->w2, cv[1].k0, cv[1].d), eT2 = dlog_of_msm(1, hT2.canon.data(), sh->c2, cv[1].k0, cv[1].d);

    reef_jacobian out;
    auto msm = [&](Curve &c, const HostVec &s, size_t n) {
        CK(reef_msm(c.key, s.mont.data(), n, REEF_HOST, true, &out, REEF_HOST));   // host scalars in, commitment back to the host: it feeds the next circuit
    };
    // warm-up (workspace allocation) and the first check
    msm(cv[0], hW1, sh->w1); check_point(cv[0], out, eW1, "comm_W1");
    msm(cv[1], hW2, sh->w2); check_point(cv[1], out, eW2, "comm_W2");

    // a config may fold in a single step: time at least three so that the per-step figure is not one sample; the totals
    // below charge the config's own number of steps
    const int timed_steps = sh->steps < 3 ? 3 : sh->steps;
    std::vector<double> step_ms;
    for (int i = 0; i < timed_steps; ++i) {
        auto ts = clk::now();
        reef_jacobian o[4];
        msm(cv[1], hT2, sh->c2); o[0] = out;
        msm(cv[0], hW1, sh->w1); o[1] = out;
        msm(cv[0], hT1, sh->c1); o[2] = out;
        msm(cv[1], hW2, sh->w2); o[3] = out;
        step_ms.push_back(ms_since(ts));
        check_point(cv[1], o[0], eT2, "comm_T2");           // outside the step's timing
        check_point(cv[0], o[1], eW1, "comm_W1");
        check_point(cv[0], o[2], eT1, "comm_T1");
        check_point(cv[1], o[3], eW2, "comm_W2");
    }
    double steps_ms = 0;
    for (double v : step_ms) steps_ms += v;
    steps_ms = steps_ms / timed_steps * sh->steps;

    // The same commitments with the two of each curve issued as ONE batched call (comm_W and comm_T of a
    // step are both absorbed before the folding challenge is drawn, so neither needs the other): rows = 2
    // over the same resident key share one pass of the pipeline.  Needs prove_step to hand both vectors
    // over together; reported next to the call-by-call figure, not instead of it.
    double steps_batched_ms = 0;
    {
        struct Pair { Curve *c; const reef_fe *a, *b; size_t len; reef_fe *buf; } pairs[2] = {
            {&cv[0], sW1, sT1, sh->w1 > sh->c1 ? sh->w1 : sh->c1, nullptr}, {&cv[1], sW2, sT2, sh->w2 > sh->c2 ? sh->w2 : sh->c2, nullptr}};
        std::vector<reef_fe> hostbuf[2];
        const HostVec *hv[2][2] = {{&hW1, &hT1}, {&hW2, &hT2}};
        reef_fe expect[2][2];
        for (int k = 0; k < 2; ++k) {
            Pair &p = pairs[k];
            hostbuf[k].resize(2 * p.len);                // the two vectors of a curve side by side in host memory
            memcpy(hostbuf[k].data(), hv[k][0]->mont.data(), p.len * sizeof(reef_fe));
            memcpy(hostbuf[k].data() + p.len, hv[k][1]->mont.data(), p.len * sizeof(reef_fe));
            for (int j = 0; j < 2; ++j) expect[k][j] = dlog_of_msm(k, hv[k][j]->canon.data(), p.len, p.c->k0, p.c->d);
        }
        reef_jacobian two[2];
        auto both = [&](int k) { CK(reef_msm_rows(pairs[k].c->key, hostbuf[k].data(), 2, pairs[k].len, REEF_HOST, true, 255, nullptr, nullptr, two, REEF_HOST)); };
        for (int k = 0; k < 2; ++k) {                    // warm-up and check
            both(k);
            check_point(*pairs[k].c, two[0], expect[k][0], "batched comm_W");
            check_point(*pairs[k].c, two[1], expect[k][1], "batched comm_T");
        }
        auto tb = clk::now();
        for (int i = 0; i < timed_steps; ++i) { both(1); both(0); }
        steps_batched_ms = ms_since(tb) / timed_steps * sh->steps;
    }

    // The pair of a curve issued as two CONCURRENT MSMs instead (reef_msm_multi: both enqueued, then both waited for; the second
    // commitment runs on a clone of the key, i.e. on another stream of the library's pool), and -- an upper bound only, because
    // the two curves of a step depend on each other through the step circuits -- all four at once.
    double steps_conc_ms = 0, steps_all4_ms = 0;
    {
        reef_msm_ctx *cl[2] = {nullptr, nullptr};
        for (int k = 0; k < 2; ++k) { CK(reef_msm_ctx_clone(&cl[k], cv[k].key)); owned_ctx.emplace_back(cl[k]); }
        reef_msm_ctx *pair_ctx[2][2] = {{cv[0].key, cl[0]}, {cv[1].key, cl[1]}};
        const reef_fe *pair_sc[2][2] = {{hW1.mont.data(), hT1.mont.data()}, {hW2.mont.data(), hT2.mont.data()}};
        const size_t pair_n[2][2] = {{sh->w1, sh->c1}, {sh->w2, sh->c2}};
        const reef_fe pair_e[2][2] = {{eW1, eT1}, {eW2, eT2}};
        reef_jacobian two[2];
        for (int k = 0; k < 2; ++k) {                    // warm-up (the clones' workspaces) and check
            CK(reef_msm_multi(2, pair_ctx[k], pair_sc[k], pair_n[k], REEF_HOST, true, two));
            check_point(cv[k], two[0], pair_e[k][0], "concurrent comm_W");
            check_point(cv[k], two[1], pair_e[k][1], "concurrent comm_T");
        }
        auto tb = clk::now();
        for (int i = 0; i < timed_steps; ++i) {
            CK(reef_msm_multi(2, pair_ctx[1], pair_sc[1], pair_n[1], REEF_HOST, true, two));
            CK(reef_msm_multi(2, pair_ctx[0], pair_sc[0], pair_n[0], REEF_HOST, true, two));
        }
        steps_conc_ms = ms_since(tb) / timed_steps * sh->steps;
        reef_msm_ctx *all_ctx[4] = {cv[1].key, cv[0].key, cl[0], cl[1]};
        const reef_fe *all_sc[4] = {hT2.mont.data(), hW1.mont.data(), hT1.mont.data(), hW2.mont.data()};
        const size_t all_n[4] = {sh->c2, sh->w1, sh->c1, sh->w2};
        reef_jacobian four[4];
        CK(reef_msm_multi(4, all_ctx, all_sc, all_n, REEF_HOST, true, four));
        check_point(cv[1], four[0], eT2, "all-four comm_T2");
        check_point(cv[0], four[2], eT1, "all-four comm_T1");
        tb = clk::now();
        for (int i = 0; i < timed_steps; ++i) CK(reef_msm_multi(4, all_ctx, all_sc, all_n, REEF_HOST, true, four));
        steps_all4_ms = ms_since(tb) / timed_steps * sh->steps;
    }

    auto t_final = clk::now();
    msm(cv[1], hT2, sh->c2);  // last NIFS fold
    int r1 = 0, r2 = 0, r3 = 0;
    const double ipa1_ms = nofold ? run_ipa_nofold(cv[0], cv[0].n, sT1, &r1) : run_ipa(cv[0], cv[0].n, sT1, &r1);
    const double ipa2_ms = nofold ? run_ipa_nofold(cv[1], cv[1].n, sT2, &r2) : run_ipa(cv[1], cv[1].n, sT2, &r2);
    const double final_ms = ms_since(t_final);

    double cons_ms = 0, concurrent_ms = 0;
    Curve hy;
    ctx_ptr hy_owner;
    if (sh->hyrax_row >= 2) {
        hy.id = REEF_PALLAS; hy.n = sh->hyrax_row; hy.d_gens = cv[0].d_gens;  // prefix of the same key shape
        for (int k = 0; k < 2; ++k) hy.ipa[k] = cv[0].ipa[k];
        if (nofold) {   // the Hyrax row generators are a resident key of their own (commitment.rs:176-186)
            reef_msm_opts o = {};
            o.bucket_groups = 1;
            o.byte_tables = tables ? 1 : 2;
            o.device = -1;
            CK(reef_msm_ctx_create(&hy.key, hy.id, hy.d_gens, hy.n, REEF_DEVICE, &o));
            hy_owner.reset(hy.key);
            reef_jacobian warm_l, warm_r;
            CK(reef_ipa_cross_terms(hy.key, sT1, hy.n, REEF_DEVICE, true, nullptr, nullptr, 0, &warm_l, &warm_r));
            cons_ms = run_ipa_nofold(hy, hy.n, sT1, &r3);
        } else {
            cons_ms = run_ipa(hy, hy.n, sT1, &r3);
        }
    }

    // The two Spartan arguments (primary and secondary curve) and the consistency argument do not depend on one another: issued
    // from three caller threads at once -- what rayon::join around them does -- their latency-bound rounds share the GPU.
    if (nofold) {
        std::exception_ptr err[2];
        auto guarded = [&](int k, Curve &c, const reef_fe *sc) {
            try { run_ipa_nofold(c, c.n, sc, nullptr); } catch (...) { err[k] = std::current_exception(); }
        };
        auto tc = clk::now();
        std::thread a(guarded, 0, std::ref(cv[0]), (const reef_fe *)sT1), b(guarded, 1, std::ref(cv[1]), (const reef_fe *)sT2);
        if (hy.key) run_ipa_nofold(hy, hy.n, sT1, nullptr);
        a.join();
        b.join();
        concurrent_ms = ms_since(tc);
        for (auto &e : err)
            if (e) std::rethrow_exception(e);
    }

    // ---- commitment of the document, the per-step sum-check and the document polynomial at proof end
    double commit_ms = 0, commit_first_ms = 0, sc_step_ms = 0, mle_ms = 0;
    if (sh->doc_log) {
        const size_t n_doc = (size_t)1 << sh->doc_log;
        const dev_ptr<uint8_t> doc_owner = device_symbols(n_doc, sh->symbol_bits, 0xD0C);
        uint8_t *d_doc = doc_owner.get();
        const size_t row_len = (size_t)1 << (sh->doc_log - sh->doc_log / 2);
        const dev_ptr<reef_affine> row_gens_owner = device_alloc<reef_affine>(row_len);
        reef_affine *d_row_gens = row_gens_owner.get();
        CK(reef_gen_bases(REEF_PALLAS, 0xFEED, 3, row_len, d_row_gens, REEF_DEVICE));
        commit_ms = run_hyrax_commit(sh, d_row_gens, d_doc, &commit_first_ms);
        std::vector<reef_fe> point(sh->doc_log);
        for (int j = 0; j < sh->doc_log; ++j) point[j] = reef_fe{{0x1f83d9abfb41bd6bULL + j, 0x5be0cd19137e2179ULL, 0x3c6ef372fe94f82bULL, 0x0a54ff53a5f1d36fULL}};
        std::vector<reef_fe> lz(row_len);
        reef_fe ev;
        CK(reef_mle_bound_rows(REEF_PALLAS, d_doc, n_doc, 1, REEF_DEVICE, true, point.data(), sh->doc_log, sh->doc_log / 2, lz.data(), REEF_HOST, &ev));
        auto t0 = clk::now();
        CK(reef_mle_bound_rows(REEF_PALLAS, d_doc, n_doc, 1, REEF_DEVICE, true, point.data(), sh->doc_log, sh->doc_log / 2, lz.data(), REEF_HOST, &ev));
        mle_ms = ms_since(t0);
    }
    if (sh->table_log) {
        const size_t len = (size_t)1 << sh->table_log;
        reef_sc_ctx *sc = nullptr;
        CK(reef_sc_create(&sc, REEF_PALLAS, len));
        const sc_ptr sc_owner(sc);
        dev_ptr<reef_fe> tab_owner = device_alloc<reef_fe>(len);
        reef_fe *d_tab = tab_owner.get();
        // the table as Reef builds it (canonical integers): --hybrid puts the transition table and then ONE value (`calc_fill`)
        // in the public half and the document in the private half (r1cs.rs:481-484, :2105-2112); otherwise it is the document
        const uint64_t sym_bound = 1ull << sh->symbol_bits;
        if (sh->table_log == sh->doc_log + 1) {
            const size_t half = len / 2, trans = std::min<size_t>(256, half / 2);
            CK(reef_gen_scalars(REEF_PALLAS, 0x7AB1E, 0, 0, trans, false, d_tab, REEF_DEVICE));
            const reef_fe fill = {{0x123456789abcdef1ULL, 0x0fedcba987654321ULL, 0x1111111122222222ULL, 0x0333333344444444ULL}};
            CK(reef_memcpy(d_tab + trans, &fill, sizeof fill, REEF_DEVICE, REEF_HOST));
            for (size_t have = 1; trans + have < half; have *= 2)
                CK(reef_memcpy(d_tab + trans + have, d_tab + trans, std::min(have, half - trans - have) * sizeof(reef_fe), REEF_DEVICE, REEF_DEVICE));
            CK(reef_gen_scalars(REEF_PALLAS, 0xD0C, 2, sym_bound, half, false, d_tab + half, REEF_DEVICE));
        } else {
            CK(reef_gen_scalars(REEF_PALLAS, 0xD0C, 2, sym_bound, len, false, d_tab, REEF_DEVICE));
        }
        CK(reef_sc_set_table(sc, 0, d_tab, len, REEF_DEVICE));
        tab_owner.reset();
        run_sumcheck_step(sc, sh->table_log, sh->lookups);            // warm-up
        // the median of three steps: one sample read 2.5 ms instead of 1.6 now and then on a box's first large run (round 4)
        double t3[3];
        for (double &t : t3) t = run_sumcheck_step(sc, sh->table_log, sh->lookups);
        std::sort(t3, t3 + 3);
        sc_step_ms = t3[1];
    }

    // ---- row N4 with STAND-IN parameters (replay_standins.h): for --merkle the Poseidon tree of the document.  Timing only: parity is
    // tests/test_gpu_merkle.py.  (Row N1, the keys derived from their labels, is the set-up above.)
    double merkle_ms = 0;
    if (sh->merkle_log) {
        const size_t n_doc = (size_t)1 << sh->merkle_log;
        std::vector<uint32_t> doc(n_doc);
        for (size_t i = 0; i < n_doc; ++i) doc[i] = (uint32_t)((i * 2654435761u) >> 24);               // one-byte symbols
        reef_poseidon_params pp;
        pp.width = 5; pp.full_rounds = STANDIN_POSEIDON_RF; pp.partial_rounds = STANDIN_POSEIDON_RP; pp.reserved = 0;
        pp.round_constants = STANDIN_POSEIDON_RC; pp.mds = STANDIN_POSEIDON_MDS;
        pp.tag_leaf = STANDIN_POSEIDON_TAGS[0]; pp.tag_node = STANDIN_POSEIDON_TAGS[1];
        reef_fe root;
        CK(reef_merkle_commit(REEF_PALLAS, &pp, doc.data(), std::min(n_doc, (size_t)1 << 16), REEF_HOST, false, nullptr, REEF_HOST, &root));   // warm-up
        auto t0 = clk::now();
        CK(reef_merkle_commit(REEF_PALLAS, &pp, doc.data(), n_doc, REEF_HOST, false, nullptr, REEF_HOST, &root));
        merkle_ms = ms_since(t0);
    }

    const std::string devices_json = ordinals.empty() ? std::string("null") : replay_devices(shape, ordinals, tables);
    const size_t pairs_step = sh->c2 + sh->w1 + sh->c1 + sh->w2;
    std::vector<char> line(16384);
    snprintf(line.data(), line.size(), "{\"replay\": \"%s\", \"ipa\": \"%s\", \"note\": \"MSM work of reef --prove replayed through the C ABI; host-side proving work not included\", "
           "\"shapes\": \"PREDICTED by Reef's cost model (src/backend/costs.rs restated in oracle/costs_oracle.py, read from %s), not measured on a Reef run\", \"w1\": %zu, \"c1\": %zu, \"w2\": %zu, \"c2\": %zu, "
           "\"scalars\": \"per-step vectors in host memory, commitments returned to the host (PCIe inclusive)\", \"commitments_checked_against_dlog\": %d, "
           "\"key_pallas\": %zu, \"key_vesta\": %zu, \"steps\": %d, \"setup_ms\": %.3f, \"setup_first_ms\": %.3f, \"setup_path\": \"both curves at once, each label -> derived generators on the device -> resident pre-shifted key (Reef re-derives its keys on every --prove: framework.rs:297-303)\", \"check_keys_ms\": %.3f, \"fold_steps_ms\": %.3f, \"ms_per_step\": %.3f, "
           "\"ms_per_step_batched_pairs\": %.3f, \"ms_per_step_concurrent\": %.3f, \"ms_per_step_all_four_at_once\": %.3f, \"pairs_per_step\": %zu, \"final_snark_ms\": %.3f, \"ipa_pallas_ms\": %.3f, \"ipa_pallas_rounds\": %d, \"ipa_vesta_ms\": %.3f, "
           "\"ipa_vesta_rounds\": %d, \"consistency_ipa_ms\": %.3f, \"consistency_rounds\": %d, \"three_arguments_concurrently_ms\": %.3f, \"total_prove_msm_ms\": %.3f, "
           "\"commit_hyrax_ms\": %.3f, \"commit_hyrax_first_call_ms\": %.3f, \"sumcheck_table_log\": %d, \"sumcheck_ms_per_step\": %.3f, "
           "\"doc_poly_bind_rows_ms\": %.3f, \"total_prove_gpu_ms\": %.3f, \"prove_gpu_incl_setup_ms\": %.3f, \"commit_merkle_log\": %d, \"commit_merkle_ms\": %.3f, "
           "\"standins\": \"key derivation and Poseidon run on stand-in parameter sets (replay_standins.h), timing only\", \"byte_tables\": %s, \"devices\": %s}",
           sh->name.c_str(), nofold ? "cross terms over the original key (no generator fold)" : "generator fold per round", shapes_path.c_str(), sh->w1, sh->c1, sh->w2, sh->c2, g_checked, cv[0].n, cv[1].n, sh->steps, setup_ms, setup_first_ms, check_keys_ms, steps_ms, steps_ms / sh->steps, steps_batched_ms / sh->steps, steps_conc_ms / sh->steps, steps_all4_ms / sh->steps, pairs_step, final_ms, ipa1_ms, r1, ipa2_ms, r2,
           cons_ms, r3, concurrent_ms, steps_ms + final_ms + cons_ms, commit_ms, commit_first_ms, sh->table_log, sc_step_ms, mle_ms,
           steps_ms + final_ms + cons_ms + sh->steps * sc_step_ms + mle_ms, setup_ms + steps_ms + final_ms + cons_ms + sh->steps * sc_step_ms + mle_ms, sh->merkle_log, merkle_ms,
           tables ? "\"built with the keys (inside setup_ms): MSMs of 1025..65536 points are sums of table entries\"" : "\"none (bucket pipeline)\"", devices_json.c_str());
    return std::string(line.data());        // the owners above release every context and device buffer, here or on an exception
}

// ---- entry points ---------------------------------------------------------------------------------------------------
// Runs the replay of the config whose name contains `config` ("cfg1" | "cfg3" | "cfg4" | "cfg5") with the shapes of
// `shapes_json` (NULL: $REEF_REPLAY_SHAPES).  On success returns 0 and writes the JSON line (NUL-terminated) to out; on
// failure returns non-zero and writes the message.  Every per-step commitment has been checked by then.
// reef_replay_run_devices: the same, followed by the multi-device leg on `devices[0 .. ndev)` (ordinals may repeat; ndev = 0: none).
extern "C" __attribute__((visibility("default")))
int reef_replay_run_devices(const char *shapes_json, const char *config, int nofold, int tables, const int *devices, size_t ndev, char *out, size_t cap) {
    auto put = [&](const std::string &m) { if (out && cap) { snprintf(out, cap, "%s", m.c_str()); } };
    try {
        std::vector<int> ordinals;
        if (ndev > 64 || (ndev && !devices)) fail("devices: 0..64 ordinals");
        for (size_t i = 0; i < ndev; ++i) {
            if (devices[i] < 0 || devices[i] >= reef_device_count()) fail("devices: ordinal " + std::to_string(devices[i]) + " is not visible");
            ordinals.push_back(devices[i]);
        }
        if (ndev && !nofold) fail("the multi-device leg replays the fold-free final SNARK: pass nofold");
        const char *path = shapes_json && *shapes_json ? shapes_json : getenv("REEF_REPLAY_SHAPES");
        if (!path) fail("no replay shapes file given (argument or REEF_REPLAY_SHAPES)");
        if (reef_device_count() < 1) { put(std::string("no GPU: ") + reef_last_error()); return 3; }
        const Shape sh = load_shape(path, config && *config ? config : "cfg3");
        g_checked = 0;
        put(replay_body(sh, nofold != 0, tables != 0, path, ordinals));
        return 0;
    } catch (const std::exception &e) {
        put(e.what());
        return 1;
    }
}
extern "C" __attribute__((visibility("default")))
int reef_replay_run(const char *shapes_json, const char *config, int nofold, int tables, char *out, size_t cap) {
    return reef_replay_run_devices(shapes_json, config, nofold, tables, nullptr, 0, out, cap);
}

#if !defined(REEF_REPLAY_NO_MAIN)
#include <unistd.h>
int main(int argc, char **argv) {
    const char *which = argc > 1 ? argv[1] : "cfg3";
    bool nofold = false, tables = false;
    std::string shapes;
    int members = 0;
    {   // an embedder's first call: eight hardware queues for the concurrent arguments, asked for before the first HIP call
        reef_runtime_opts ro = {};
        ro.hw_queues = 8;
        (void)reef_runtime_init(&ro, nullptr);
    }
    for (int i = 2; i < argc; ++i) {
        if (strcmp(argv[i], "nofold") == 0) nofold = true;
        else if (strcmp(argv[i], "tables") == 0) tables = true;       // the keys' byte tables are ready before the first MSM (a real run builds them in the background)
        else if (strncmp(argv[i], "shapes=", 7) == 0) shapes = argv[i] + 7;
        else if (strncmp(argv[i], "devices=", 8) == 0) members = atoi(argv[i] + 8);     // the multi-device leg on N members (ordinal i mod the visible devices)
        else { fprintf(stderr, "usage: reef_replay [cfg1|cfg3|cfg4|cfg5] [nofold] [tables] [devices=N] [shapes=<replay_shapes.json>]\n"); return 2; }
    }
    if (shapes.empty() && !getenv("REEF_REPLAY_SHAPES")) {   // the executable lives in reef_amd/_lib/: the shapes are two levels up, under tests/golden/
        char exe[4096];
        const ssize_t len = readlink("/proc/self/exe", exe, sizeof exe - 1);
        if (len > 0) {
            exe[len] = 0;
            std::string dir(exe);
            dir = dir.substr(0, dir.rfind('/'));
            shapes = dir + "/../../tests/golden/replay_shapes.json";
        }
    }
    std::vector<char> out(32768);
    std::vector<int> ordinals;
    const int visible = reef_device_count();
    for (int i = 0; i < members && visible > 0; ++i) ordinals.push_back(i % visible);
    const int rc = reef_replay_run_devices(shapes.empty() ? nullptr : shapes.c_str(), which, nofold, tables, ordinals.data(), ordinals.size(), out.data(), out.size());
    if (rc == 0) printf("%s\n", out.data());
    else fprintf(stderr, "reef_replay: %s\n", out.data());
    return rc;
}
#endif
