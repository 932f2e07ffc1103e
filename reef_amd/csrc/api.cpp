// C ABI of libreef_msm.so (see include/reef_msm.h).  Thin dispatch onto the per-curve engines;
// no arithmetic here and no CPU fallback: without a gfx950 device every call fails loudly.
#include <stdarg.h>
#include <stdio.h>
#include <string>
#include <vector>
#include <new>
#include <exception>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

#include "common.h"
#include "keccak.h"

namespace reef {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless set), and streams that share
// a queue run one after the other.  One caller thread is a context stream plus its scratch stream: with the three arguments of
// the final SNARK issued at once that is more than four, and their latency-bound rounds queue up behind each
// other -- 8.3 ms against 6.4 ms with eight queues for cfg3, 4 threads of IPA rounds 1.95x against 2.55x one thread
// (tools/time_concurrent_ipa.py).  The variable is read when the runtime initialises, i.e. at the process's first HIP call.
// OPT-IN since round 5 (rounds 3-4 put 8 into the environment whenever the library was loaded): an embedder
// calls reef_runtime_init({hw_queues = 8}) before its first HIP call (explicit, reported back), or exports REEF_MSM_HW_QUEUES=<n>
// -- then the library asks for n when it is loaded.  Without either the runtime is left alone.  A value the user exported as
// GPU_MAX_HW_QUEUES always wins, and the library says once on stderr when it has put the variable there.
namespace {
std::atomic<int> g_hw_queues_asked{0};     // what this library put into the environment (0: nothing)
static int ask_hw_queues(int n) {
    if (n <= 0) return 0;
    if (getenv("GPU_MAX_HW_QUEUES")) return 0;          // the user's (or an earlier call's) value stands
    char buf[16];
    snprintf(buf, sizeof buf, "%d", n);
    setenv("GPU_MAX_HW_QUEUES", buf, 0);
    g_hw_queues_asked.store(n);
    static std::atomic<bool> said{false};
    const char *l = getenv("REEF_MSM_LOG");
    if (!said.exchange(true) && !(l && l[0] == '0' && l[1] == 0))
        fprintf(stderr, "libreef_msm: GPU_MAX_HW_QUEUES=%d set for this process on request (reef_runtime_init / REEF_MSM_HW_QUEUES; REEF_MSM_LOG=0 silences this line)\n", n);
    return n;
}
struct HwQueues {
    HwQueues() {
        const char *o = getenv("REEF_MSM_HW_QUEUES");
        if (o && *o && atoi(o) > 0) ask_hw_queues(atoi(o));   // only on the user's explicit request
    }
} g_hw_queues;
}  // namespace

// Thread-local destructors of the main thread and static destructors run at process teardown, when HIP may be gone: they look
// at this flag (set by an atexit hook) and leave the memory to the driver.
static std::atomic<bool> g_process_exiting{false};
static std::atomic<int> g_warm_state{0};     // reef_runtime_init's warm-up: 0 never asked, 1 running, 2 done, 3 failed
static void hook_process_exit() {
    static std::once_flag once;
    std::call_once(once, [] {
        atexit([] {
            g_process_exiting.store(true);
            // a warm-up thread still inside the runtime's initialisation must not meet the runtime's teardown (bounded: it takes ~100 ms)
            for (int i = 0; i < 4000 && g_warm_state.load() == 1; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        });
    });
}

static const CurveVTable *vt(int curve) {
    if (curve == REEF_PALLAS) return pallas_vtable();
    if (curve == REEF_VESTA) return vesta_vtable();
    set_error("unknown curve %d", curve);
    return nullptr;
}

static reef_status require_gpu() {
    static std::once_flag once;
    static reef_status st = REEF_OK;
    static char msg[256];
    std::call_once(once, [] {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n == 0) {
            snprintf(msg, sizeof msg, "no HIP device visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
            st = REEF_ERR_NO_GPU;
            return;
        }
        hipDeviceProp_t prop;
        int dev = 0;
        (void)hipGetDevice(&dev);
        e = hipGetDeviceProperties(&prop, dev);
        if (e != hipSuccess) { snprintf(msg, sizeof msg, "hipGetDeviceProperties: %s", hipGetErrorString(e)); st = REEF_ERR_HIP; return; }
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            snprintf(msg, sizeof msg, "device %d is %s; libreef_msm.so carries gfx950 code only", dev, prop.gcnArchName);
            st = REEF_ERR_NO_GPU;
        }
    });
    if (st != REEF_OK) set_error("%s", msg);
    return st;
}

// The runtime's one-off costs, paid where nobody waits for them (include/reef_msm.h: reef_runtime_opts.warm): runtime initialisation, the pool's first
// stream, one launch out of each curve's code object, a copy in each direction from and to pageable memory -- what the first commitment of a process
// otherwise pays (~100 ms, profiles/r06_seam_hip_first_use.txt).
static void warm_body() {
    if (require_gpu() != REEF_OK) { g_warm_state.store(3); return; }
    reef_fe a[2], b[2], out[2];
    memset(a, 0, sizeof a);
    memset(b, 0, sizeof b);
    a[0].l[0] = 5; a[1].l[0] = 7; b[0].l[0] = 3; b[1].l[0] = 11;
    bool ok = true;
    for (int curve = 0; curve < 2; ++curve)      // stage_in (pageable host -> device), a kernel of this curve's code object, device -> pageable host
        ok = ok && vt(curve)->test_field_op(0, a, b, out, 2) == REEF_OK;
    // every kernel of the bucket pipeline once per curve (a kernel's first launch resolves it in its code object), the engine's one-off attributes, the first
    // pinned allocation: a 2048-point MSM over identity points on a plain key, host scalars in, host result out -- and gone again
    {
        const size_t n = 2048;
        std::vector<reef_affine> pts(n);
        std::vector<reef_fe> sc(n);
        memset(pts.data(), 0, n * sizeof(reef_affine));                  // (0, 0): the identity
        for (size_t i = 0; i < n; ++i) { sc[i].l[0] = 0x9e3779b97f4a7c15ull * (i + 1); sc[i].l[1] = i; sc[i].l[2] = ~i; sc[i].l[3] = 0x0123456789abcdefull >> 3; }
        for (int curve = 0; curve < 2 && ok; ++curve) {
            PoolNoGrowth ng;
            reef_msm_ctx *c = nullptr;
            reef_jacobian out1;
            ok = reef_msm_ctx_create(&c, curve, pts.data(), n, REEF_HOST, nullptr) == REEF_OK;
            ok = ok && reef_msm(c, sc.data(), n, REEF_HOST, false, &out1, REEF_HOST) == REEF_OK;
            reef_msm_ctx_destroy(c);
        }
    }
    // the builder thread's own stream (keys of 2^17 points and more are built there, common.h): its creation is 7-9 ms of the first such key's build otherwise
    int dev = 0;
    if (ok && hipGetDevice(&dev) == hipSuccess) (void)stream_pool().ensure_background_stream(dev);
    g_warm_state.store(ok ? 2 : 3);
}
static void start_warm(uint32_t mode) {
    if (mode == REEF_WARM_NONE) return;
    int expect = 0;
    if (!g_warm_state.compare_exchange_strong(expect, 1)) return;     // once per process
    hook_process_exit();
    if (mode == REEF_WARM_NOW) { warm_body(); return; }
    std::thread([] { warm_body(); }).detach();
}
namespace {
struct WarmAtLoad {                                  // the zero-patch route has no init call: REEF_MSM_WARM=1 in the environment asks when the library is loaded
    WarmAtLoad() {
        const char *e = getenv("REEF_MSM_WARM");
        if (e && *e && atoi(e) > 0) start_warm(REEF_WARM_BACKGROUND);   // (its first step is the runtime's own initialisation, tens of ms: the loader has long finished this library's other constructors, the code objects' registration among them)
    }
} g_warm_at_load;
}  // namespace

}  // namespace reef

using namespace reef;

struct reef_msm_ctx {
    int curve;
    void *impl;
};

// No C++ exception crosses the C ABI: host allocations sized by the caller (key-derivation stream, staging vectors) can fail.
template <class F> static reef_status guarded(F &&f) {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        set_error("host memory exhausted");
        return REEF_ERR_OOM;
    } catch (const std::exception &e) {
        set_error("internal error: %s", e.what());
        return REEF_ERR_HIP;
    }
}

extern "C" {

const char *reef_last_error(void) { return g_err; }
const char *reef_version(void) {
#ifdef REEF_EXPERIMENT
    return "reef_msm 0.6 (gfx950; +experiment: the A/B switches of common.h are compiled in)";
#else
    return "reef_msm 0.6 (gfx950; release)";
#endif
}
uint32_t reef_abi_version(void) { return REEF_ABI_VERSION; }
reef_status reef_runtime_init(const reef_runtime_opts *opts, reef_runtime_info *info) {
    if (opts && opts->hw_queues > 0) {
        // an explicit request replaces what the library constructor put there (never what the user exported)
        if (g_hw_queues_asked.load() > 0) unsetenv("GPU_MAX_HW_QUEUES");
        ask_hw_queues(opts->hw_queues);
    } else if (opts && opts->hw_queues < 0 && g_hw_queues_asked.load() > 0) {
        unsetenv("GPU_MAX_HW_QUEUES");               // leave the runtime alone: take the constructor's value back
        g_hw_queues_asked.store(0);
    }
    if (opts && opts->warm) {
        if (opts->warm > REEF_WARM_BACKGROUND) { set_error("reef_runtime_init: unknown warm mode %u", opts->warm); return REEF_ERR_ARG; }
        start_warm(opts->warm);                      // after the hardware queues have been asked for: the warm-up is the process's first HIP call
    }
    if (info) {
        const char *e = getenv("GPU_MAX_HW_QUEUES");
        info->hw_queues_env = e ? atoi(e) : 0;
        info->hw_queues_set_by_library = g_hw_queues_asked.load();
        info->abi_version = REEF_ABI_VERSION;
        info->warm = (uint32_t)g_warm_state.load();
    }
    return REEF_OK;
}

int reef_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
reef_status reef_set_device(int ordinal) {
    REEF_HIP_TRY(hipSetDevice(ordinal));
    return REEF_OK;
}
reef_status reef_get_device(int *ordinal) {
    if (!ordinal) { set_error("null argument"); return REEF_ERR_ARG; }
    REEF_HIP_TRY(hipGetDevice(ordinal));
    return REEF_OK;
}
reef_status reef_device_sync(void) {
    REEF_HIP_TRY(hipDeviceSynchronize());
    return REEF_OK;
}
void *reef_device_alloc(size_t bytes) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 16);
    if (e != hipSuccess) { set_error("hipMalloc(%zu): %s", bytes, hipGetErrorString(e)); return nullptr; }
    return p;
}
void reef_device_free(void *p) {
    if (p) (void)hipFree(p);
}
reef_status reef_memcpy(void *dst, const void *src, size_t bytes, int dst_loc, int src_loc) {
    hipMemcpyKind k = dst_loc == REEF_DEVICE ? (src_loc == REEF_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice)
                                             : (src_loc == REEF_DEVICE ? hipMemcpyDeviceToHost : hipMemcpyHostToHost);
    REEF_HIP_TRY(hipMemcpy(dst, src, bytes, k));
    return REEF_OK;
}

reef_status reef_msm_ctx_create(reef_msm_ctx **out, int curve, const reef_affine *bases, size_t n, int bases_loc,
                                const reef_msm_opts *opts) {
    if (!out) { set_error("null argument"); return REEF_ERR_ARG; }
    const CurveVTable *v = vt(curve);
    if (!v) return REEF_ERR_ARG;
    REEF_TRY(require_gpu());
    void *impl = nullptr;
    REEF_TRY(v->ctx_create(&impl, bases, n, bases_loc, opts));
    *out = new reef_msm_ctx{curve, impl};
    return REEF_OK;
}
reef_status reef_msm_ctx_set_bases(reef_msm_ctx *ctx, const reef_affine *bases, size_t n, int bases_loc) {
    if (!ctx || (n && !bases)) { set_error("null argument"); return REEF_ERR_ARG; }
    return vt(ctx->curve)->ctx_rekey(ctx->impl, bases, n, bases_loc);
}
reef_status reef_msm_ctx_attach(reef_msm_ctx *ctx, reef_msm_ctx *src) {
    if (!ctx || !src || ctx->curve != src->curve) { set_error("attach: two contexts of the same curve"); return REEF_ERR_ARG; }
    return vt(ctx->curve)->ctx_attach(ctx->impl, src->impl);
}
reef_status reef_msm_ctx_clone(reef_msm_ctx **out, reef_msm_ctx *src) {
    if (!out || !src) { set_error("null argument"); return REEF_ERR_ARG; }
    void *impl = nullptr;
    REEF_TRY(vt(src->curve)->ctx_clone(&impl, src->impl));
    *out = new reef_msm_ctx{src->curve, impl};
    return REEF_OK;
}
void reef_msm_ctx_destroy(reef_msm_ctx *ctx) {
    if (!ctx) return;
    vt(ctx->curve)->ctx_destroy(ctx->impl);
    delete ctx;
}
reef_status reef_msm_ctx_sync(reef_msm_ctx *ctx) {
    if (!ctx) { set_error("null argument"); return REEF_ERR_ARG; }
    return vt(ctx->curve)->ctx_sync(ctx->impl);
}
void *reef_msm_ctx_stream(reef_msm_ctx *ctx) { return ctx ? vt(ctx->curve)->ctx_stream(ctx->impl) : nullptr; }
reef_status reef_msm_ctx_last_timing(reef_msm_ctx *ctx, float *total_ms, float *accumulate_ms) {
    if (!ctx) { set_error("null argument"); return REEF_ERR_ARG; }
    return vt(ctx->curve)->ctx_timing(ctx->impl, total_ms, accumulate_ms);
}
reef_status reef_msm_ctx_set_window_split(reef_msm_ctx *ctx, uint32_t rank, uint32_t world) {
    if (!ctx) { set_error("null argument"); return REEF_ERR_ARG; }
    return vt(ctx->curve)->ctx_window_split(ctx->impl, rank, world);
}
reef_status reef_msm_ctx_enable_timing(reef_msm_ctx *ctx, int on) {
    if (!ctx) { set_error("null argument"); return REEF_ERR_ARG; }
    return vt(ctx->curve)->ctx_enable_timing(ctx->impl, on);
}
reef_status reef_msm_ctx_timing_stats(reef_msm_ctx *ctx, int reset, uint64_t *calls, double *total_ms, double *accumulate_ms) {
    if (!ctx) { set_error("null argument"); return REEF_ERR_ARG; }
    return vt(ctx->curve)->ctx_timing_stats(ctx->impl, reset, calls, total_ms, accumulate_ms);
}
reef_status reef_msm_ctx_sum_points(reef_msm_ctx *ctx, const reef_jacobian *in, size_t n, reef_jacobian *out) {
    if (!ctx) { set_error("null argument"); return REEF_ERR_ARG; }
    return vt(ctx->curve)->ctx_sum_points(ctx->impl, in, n, out);
}
reef_status reef_msm_ctx_plan(reef_msm_ctx *ctx, uint32_t *c, uint32_t *windows, uint32_t *groups, uint32_t *tables) {
    if (!ctx) { set_error("null argument"); return REEF_ERR_ARG; }
    return vt(ctx->curve)->ctx_plan(ctx->impl, c, windows, groups, tables);
}

reef_status reef_msm(reef_msm_ctx *ctx, const reef_fe *scalars, size_t n, int scalars_loc, bool is_mont, reef_jacobian *out,
                     int out_loc) {
    if (!ctx) { set_error("null argument"); return REEF_ERR_ARG; }
    return guarded([&] { return vt(ctx->curve)->msm(ctx->impl, scalars, n, scalars_loc, is_mont, out, out_loc); });
}
reef_status reef_msm_rows(reef_msm_ctx *ctx, const reef_fe *scalars, size_t rows, size_t row_len, int scalars_loc, bool is_mont,
                          uint32_t max_scalar_bits, const reef_fe *blinds, const reef_affine *h, reef_jacobian *out, int out_loc) {
    if (!ctx) { set_error("null argument"); return REEF_ERR_ARG; }
    return guarded([&] { return vt(ctx->curve)->msm_rows(ctx->impl, scalars, rows, row_len, scalars_loc, is_mont, max_scalar_bits, blinds, h, out, out_loc); });
}

// Several MSMs at once, each on its own context: everything is enqueued first (a context with work in flight keeps its stream,
// so the next one takes another stream of the pool), then the results are waited for.  The commitments land in host-mapped
// memory of the calling thread: one wait per context, no copy kernel.
reef_status reef_msm_multi(size_t count, reef_msm_ctx *const *ctxs, const reef_fe *const *scalars, const size_t *n, int scalars_loc, bool is_mont,
                           reef_jacobian *out) {
    if (count == 0) return REEF_OK;
    if (!ctxs || !scalars || !n || !out || count > 64) { set_error("reef_msm_multi: null argument or more than 64 MSMs"); return REEF_ERR_ARG; }
    for (size_t i = 0; i < count; ++i) {
        if (!ctxs[i]) { set_error("null argument"); return REEF_ERR_ARG; }
        for (size_t j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) { set_error("reef_msm_multi: contexts must be distinct (clone a key that is used twice)"); return REEF_ERR_ARG; }
    }
    static thread_local struct Landing {
        reef_jacobian *p = nullptr;
        ~Landing() { if (p && !g_process_exiting.load()) (void)hipHostFree(p); }
    } land;
    hook_process_exit();
    if (!land.p) REEF_HIP_TRY(hipHostMalloc((void **)&land.p, 64 * sizeof(reef_jacobian), hipHostMallocDefault));
    return guarded([&] {
        reef_status st = REEF_OK;
        size_t issued = 0;
        for (; issued < count && st == REEF_OK; ++issued)
            st = vt(ctxs[issued]->curve)->msm(ctxs[issued]->impl, scalars[issued], n[issued], scalars_loc, is_mont, land.p + issued, REEF_DEVICE);
        for (size_t i = 0; i < issued; ++i) {           // wait for whatever was enqueued, also after a failure
            const reef_status w = vt(ctxs[i]->curve)->ctx_sync(ctxs[i]->impl);
            if (st == REEF_OK) st = w;
        }
        if (st == REEF_OK) memcpy(out, land.p, count * sizeof(reef_jacobian));
        return st;
    });
}

int reef_msm_ctx_byte_tables(reef_msm_ctx *ctx) { return ctx ? vt(ctx->curve)->ctx_byte_tables(ctx->impl) : 0; }
reef_status reef_msm_plan_for(size_t n, uint32_t window_bits, uint32_t bucket_groups, uint32_t *c, uint32_t *windows,
                              uint32_t *groups, uint32_t *tables) {
    return pallas_vtable()->plan_for(n, window_bits, bucket_groups, c, windows, groups, tables);
}

reef_status reef_msm_rows_symbols(reef_msm_ctx *ctx, const uint8_t *symbols, size_t rows, size_t row_len, int loc, uint32_t symbol_bits,
                                  const reef_fe *blinds, const reef_affine *h, bool blinds_are_mont, reef_jacobian *out, int out_loc) {
    if (!ctx) { set_error("null argument"); return REEF_ERR_ARG; }
    return guarded([&] { return vt(ctx->curve)->msm_rows_symbols(ctx->impl, symbols, rows, row_len, loc, symbol_bits, blinds, h, blinds_are_mont, out, out_loc); });
}
reef_status reef_ipa_cross_terms(reef_msm_ctx *ctx, const reef_fe *a, size_t n_k, int a_loc, bool is_mont, const reef_fe *w1s,
                                 const reef_fe *w2s, size_t k, reef_jacobian *out_l, reef_jacobian *out_r) {
    if (!ctx) { set_error("null argument"); return REEF_ERR_ARG; }
    return guarded([&] { return vt(ctx->curve)->ipa_cross(ctx->impl, a, n_k, a_loc, is_mont, w1s, w2s, k, out_l, out_r); });
}

reef_status reef_msm_folded(reef_msm_ctx *ctx, const reef_fe *v, size_t len, size_t off, int v_loc, bool is_mont, const reef_fe *w1s,
                            const reef_fe *w2s, size_t k, reef_jacobian *out, int out_loc) {
    if (!ctx) { set_error("null argument"); return REEF_ERR_ARG; }
    return guarded([&] { return vt(ctx->curve)->msm_folded(ctx->impl, v, len, off, v_loc, is_mont, w1s, w2s, k, out, out_loc); });
}

#define STATELESS_PROLOGUE(curve)          \
    const CurveVTable *v = vt(curve);      \
    if (!v) return REEF_ERR_ARG;           \
    REEF_TRY(require_gpu())

reef_status reef_fold(int curve, const reef_affine *gens, size_t half, int loc, const reef_fe *w1, const reef_fe *w2,
                      reef_affine *out) {
    STATELESS_PROLOGUE(curve);
    return v->fold(gens, half, loc, w1, w2, out);
}
reef_status reef_mle_bound_rows(int curve, const void *z, size_t n, int elem_bytes, int z_loc, bool is_mont, const reef_fe *point,
                                size_t num_vars, size_t left_vars, reef_fe *lz_out, int out_loc, reef_fe *eval_out) {
    STATELESS_PROLOGUE(curve);
    return v->mle_bound(z, n, elem_bytes, z_loc, is_mont, point, num_vars, left_vars, lz_out, out_loc, eval_out);
}
reef_status reef_normalize(int curve, const reef_jacobian *in, size_t n, int loc, reef_affine *out_affine, uint8_t *out_compressed) {
    STATELESS_PROLOGUE(curve);
    return v->normalize(in, n, loc, out_affine, out_compressed);
}
reef_status reef_sum_points(int curve, const reef_jacobian *in, size_t n, int loc, reef_jacobian *out) {
    STATELESS_PROLOGUE(curve);
    return v->sum_points(in, n, loc, out);
}
reef_status reef_gen_bases(int curve, uint64_t k0, uint64_t d, size_t n, reef_affine *out, int loc) {
    STATELESS_PROLOGUE(curve);
    return v->gen_bases(k0, d, n, out, loc);
}
reef_status reef_gen_scalars(int curve, uint64_t seed, int kind, uint64_t small_bound, size_t n, bool to_mont, reef_fe *out,
                             int loc) {
    STATELESS_PROLOGUE(curve);
    return v->gen_scalars(seed, kind, small_bound, n, to_mont, out, loc);
}
reef_status reef_test_field_op(int field, int op, const reef_fe *a, const reef_fe *b, reef_fe *out, size_t n) {
    STATELESS_PROLOGUE(field);  // coordinate field of curve `field`
    return v->test_field_op(op, a, b, out, n);
}
reef_status reef_test_ec_op(int curve, int op, const reef_affine *p, const reef_affine *q, const reef_fe *k, reef_jacobian *out,
                            size_t n) {
    STATELESS_PROLOGUE(curve);
    return v->test_ec_op(op, p, q, k, out, n);
}
reef_status reef_bench_fmul(int field, uint32_t iters, double *products_per_s) {
    STATELESS_PROLOGUE(field);
    if (!products_per_s) { set_error("null argument"); return REEF_ERR_ARG; }
    return v->bench_fmul(iters, products_per_s);
}

// ---- row N2: sum-check vector kernels
struct reef_sc_ctx {
    int curve;
    void *impl;
};
reef_status reef_sc_create(reef_sc_ctx **out, int curve, size_t table_len) {
    if (!out) { set_error("null argument"); return REEF_ERR_ARG; }
    STATELESS_PROLOGUE(curve);
    void *impl = nullptr;
    REEF_TRY(v->sc_create(&impl, table_len));
    *out = new reef_sc_ctx{curve, impl};
    return REEF_OK;
}
void reef_sc_destroy(reef_sc_ctx *ctx) {
    if (!ctx) return;
    vt(ctx->curve)->sc_destroy(ctx->impl);
    delete ctx;
}
#define SC_CHECK(ctx) if (!(ctx)) { set_error("null argument"); return REEF_ERR_ARG; }
reef_status reef_sc_set_table(reef_sc_ctx *ctx, int which, const reef_fe *values, size_t n, int loc) {
    SC_CHECK(ctx);
    return vt(ctx->curve)->sc_set(ctx->impl, which, values, n, loc);
}
reef_status reef_sc_gen_eq_table(reef_sc_ctx *ctx, const reef_fe *rs, const uint32_t *qs, size_t nq, const reef_fe *last_q, size_t ell) {
    SC_CHECK(ctx);
    return vt(ctx->curve)->sc_gen_eq(ctx->impl, rs, qs, nq, last_q, ell);
}
reef_status reef_sc_round_coeffs(reef_sc_ctx *ctx, size_t pow, reef_fe out[3]) {
    SC_CHECK(ctx);
    return vt(ctx->curve)->sc_coeffs(ctx->impl, pow, out);
}
reef_status reef_sc_fold(reef_sc_ctx *ctx, size_t pow, const reef_fe *r) {
    SC_CHECK(ctx);
    return vt(ctx->curve)->sc_fold(ctx->impl, pow, r);
}
reef_status reef_sc_fold_and_next_coeffs(reef_sc_ctx *ctx, size_t pow, const reef_fe *r, reef_fe out[3]) {
    SC_CHECK(ctx);
    return vt(ctx->curve)->sc_fold_coeffs(ctx->impl, pow, r, out);
}
reef_status reef_sc_read(reef_sc_ctx *ctx, int which, size_t count, reef_fe *out) {
    SC_CHECK(ctx);
    return vt(ctx->curve)->sc_read(ctx->impl, which, count, out);
}
reef_status reef_sc_reset_table(reef_sc_ctx *ctx) {
    SC_CHECK(ctx);
    return vt(ctx->curve)->sc_reset(ctx->impl);
}
reef_status reef_sc_sync(reef_sc_ctx *ctx) {
    SC_CHECK(ctx);
    return vt(ctx->curve)->sc_sync(ctx->impl);
}

uint64_t reef_merkle_nodes(uint64_t n) {
    uint64_t total = 0, m = (n + 1) / 2;
    for (;;) {
        total += m;
        if (m <= 1) break;
        m = (m + 1) / 2;
    }
    return total;
}
reef_status reef_merkle_commit(int curve, const reef_poseidon_params *params, const uint32_t *doc, size_t n, int doc_loc, bool is_mont,
                               reef_fe *tree_out, int tree_loc, reef_fe *root_out) {
    const CurveVTable *v = vt(curve);
    if (!v) return REEF_ERR_ARG;
    REEF_TRY(require_gpu());
    return guarded([&] { return v->merkle_commit(params, doc, n, doc_loc, is_mont, tree_out, tree_loc, root_out, nullptr); });
}

// The same tree built by several devices of this process (include/reef_msm.h 3d).  The bottom level is cut into blocks of S = 2^L nodes, one
// block per device; a block is a subtree of the whole tree, so the devices exchange nothing but their block's root (32 bytes each, through the
// host), and the levels above L are hashed from those roots on devices[0].
reef_status reef_merkle_commit_devices(int curve, const reef_poseidon_params *params, const uint32_t *doc, size_t n, bool is_mont, const int *devices,
                                       size_t ndev, reef_fe *tree_out, reef_fe *root_out, uint32_t *blocks_out) {
    const CurveVTable *v = vt(curve);
    if (!v) return REEF_ERR_ARG;
    if (!devices || ndev == 0 || ndev > 64) { set_error("reef_merkle_commit_devices: devices"); return REEF_ERR_ARG; }
    if (!params || (n && !doc) || (!tree_out && !root_out)) { set_error("null argument"); return REEF_ERR_ARG; }
    if (n == 0 || n >= (1ull << 33)) { set_error("document length out of range"); return REEF_ERR_ARG; }
    REEF_TRY(require_gpu());
    const int visible = reef_device_count();
    for (size_t i = 0; i < ndev; ++i)
        if (devices[i] < 0 || devices[i] >= visible) { set_error("reef_merkle_commit_devices: devices[%zu] = %d, %d visible", i, devices[i], visible); return REEF_ERR_ARG; }
    return guarded([&]() -> reef_status {
        // global shape of the tree: sizes and offsets of its levels (merkle_tree.rs:25-80)
        std::vector<uint64_t> size, off;
        {
            uint64_t m = ((uint64_t)n + 1) / 2, o = 0;
            for (;;) {
                size.push_back(m); off.push_back(o);
                o += m;
                if (m <= 1) break;
                m = (m + 1) / 2;
            }
        }
        uint32_t L = 0;                                    // block height: the smallest power of two with at most ndev blocks
        while (((size[0] + (1ull << L) - 1) >> L) > ndev) ++L;
        const uint64_t S = 1ull << L;
        const size_t nb = (size_t)((size[0] + S - 1) / S);
        if (blocks_out) *blocks_out = (uint32_t)nb;
        if (nb == 1) {
            REEF_ON_DEVICE(devices[0]);
            return v->merkle_commit(params, doc, n, REEF_HOST, is_mont, tree_out, REEF_HOST, root_out, nullptr);
        }
        std::vector<reef_fe> roots(nb);
        std::vector<reef_status> st(nb, REEF_OK);
        std::vector<std::string> msg(nb);
        std::vector<std::thread> th;
        for (size_t b = 0; b < nb; ++b)
            th.emplace_back([&, b] {
                DeviceGuard ds(devices[b]);
                if (!ds.ok) { set_error("cannot make device %d current: %s", devices[b], hipGetErrorString(ds.err)); st[b] = REEF_ERR_HIP; msg[b] = reef_last_error(); return; }
                MerkleSlice sl;
                sl.index_base = 2 * b * S;
                sl.levels = L;
                std::vector<reef_fe *> dst(L + 1, nullptr);
                if (tree_out)
                    for (uint32_t h = 0; h <= L; ++h) dst[h] = tree_out + off[h] + ((b * S) >> h);
                sl.level_out = dst.data();
                sl.top_out = &roots[b];
                const uint64_t first = 2 * b * S, cnt = std::min<uint64_t>((uint64_t)n - first, 2 * S);
                try {
                    st[b] = v->merkle_commit(params, doc + first, (size_t)cnt, REEF_HOST, is_mont, nullptr, REEF_HOST, nullptr, &sl);
                } catch (const std::exception &e) {
                    set_error("internal error: %s", e.what());
                    st[b] = REEF_ERR_HIP;
                }
                if (st[b] != REEF_OK) msg[b] = reef_last_error();
            });
        for (auto &t : th) t.join();
        for (size_t b = 0; b < nb; ++b)
            if (st[b] != REEF_OK) { set_error("block %zu (device %d): %s", b, devices[b], msg[b].c_str()); return st[b]; }
        // the levels above the blocks, from their roots
        MerkleSlice top;
        top.nodes_in = roots.data();
        top.nodes_n = nb;
        top.levels = (uint32_t)(size.size() - 1 - L);
        std::vector<reef_fe *> dst(top.levels + 1, nullptr);
        if (tree_out)
            for (uint32_t h = 1; h <= top.levels; ++h) dst[h] = tree_out + off[L + h];    // level L itself is the blocks' own
        top.level_out = dst.data();
        reef_fe root;
        top.top_out = &root;
        REEF_ON_DEVICE(devices[0]);
        REEF_TRY(v->merkle_commit(params, nullptr, 0, REEF_HOST, is_mont, nullptr, REEF_HOST, nullptr, &top));
        if (root_out) *root_out = root;
        return REEF_OK;
    });
}

reef_status reef_derive_generators(int curve, const uint8_t *label, size_t label_len, size_t n, const reef_keygen_params *params, bool is_mont,
                                   reef_affine *out, int out_loc) {
    const CurveVTable *v = vt(curve);
    if (!v) return REEF_ERR_ARG;
    REEF_TRY(require_gpu());
    return guarded([&] { return v->derive_generators(label, label_len, n, params, is_mont, out, out_loc); });
}
void reef_shake256(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) { reef::shake256(in, in_len, out, out_len); }

// ---- pasta-msm drop-in symbols: stateless, abort on failure (the Rust side panics on error).
// The bases are read on every call, as the reference semantics (nothing retained that the caller can observe) require.
namespace {
// A commitment key that keeps coming back (Reef commits to the same generators in every folding step,
// src/backend/framework.rs:297-303) is recognised by its bytes; once a resident pre-shifted copy of it exists the call runs on that
// copy (import, the plain-key pipeline and the host-side window combine are skipped).  Keys seen once -- the folded generators of an
// IPA round -- never get a resident copy.  REEF_MSM_KEY_CACHE=0 turns it off.
//
// Hashes only NOMINATE an entry; a hit is CONFIRMED by comparing every byte the caller passed with the copy retained next to
// the resident key, so a collision costs time, never a wrong commitment.  The confirmation is free: the caller's thread enqueues
// the MSM on the nominated key FIRST (speculation) and compares the bytes while the GPU works -- always on the HOST, against a host
// copy (round 5: no base crosses PCIe on a hit, whatever the key's size).  Keys of more than 4 MiB are compared in 1 MiB pieces by
// the caller and a few helper threads of the process (REEF_MSM_CMP_THREADS, default up to 8: 64 MiB in ~1 ms, beside the scalars'
// upload and the MSM).  The result is taken only if every byte was equal; otherwise the call is served again on the plain path.
//
// Round 6 -- NOTHING of a key's warm-up runs on the caller's critical path any more (the call that brought a key's second
// appearance built the resident copy synchronously, 10-49 ms, and the third paid a context of the thread's own, up to 8 ms; a Reef proof
// folds 1-6 times, so the warm-up WAS the cost).  The call that brings the second appearance is served on the plain path like the first,
// keeps a copy of the key's bytes (a memcpy; pieces shared with the helper threads above 4 MiB) and returns; ONE builder thread of the
// process (Builder, below) uploads that copy, builds the pre-shifted tables -- on a stream the callers already use: creating one would stall them (common.h) -- and, before it publishes the key,
// prepares a spare context whose workspace has already served a (zero-scalar) MSM of the key's length, with its host-mapped landing zone.
// The first call that finds the key published takes that spare instead of creating a context.  Until then calls keep to the plain path:
// a caller never waits for the builder thread (its kernels are ordered among the caller's own on the shared stream: one call waits the 1-2 ms of
// GPU time the tables take).  The builder also keeps the contexts threads let go of as spares (a pool of three per curve; destroying one
// beside a busy caller is thirty hipFree waits), each still holding its key until it is attached elsewhere or destroyed.
//
// The table of resident keys is ONE per process (nova-snark reaches this symbol from the prover thread and from rayon workers,
// src/backend/framework.rs:110,668,695): a key is built once, and every caller thread serves it through a context of its own attached
// to the key of the moment (reef_msm_ctx_attach: O(1)).  The table lock is held for lookups only, NEVER across HIP work or a destructor
// that does HIP work: entries that leave the table are destroyed after the lock is released, and an entry lives on --
// charged to the budget -- until the last context attached to it (a thread's, or a pooled spare) has moved on or ended.  Device memory is charged to one
// budget (REEF_MSM_KEY_CACHE_MB, default 16384), the host copies to another (REEF_MSM_KEY_HOST_MB, default 4096), both RESERVED
// atomically before anything is allocated; a key that finds a budget full goes back to "nominated" and is tried again when it
// returns.  An allocation failure anywhere on this path empties the table, tells every thread to let go of its attachment at its next
// call (an epoch), and retries once on the plain, uncached path before the symbol gives up.
struct SharedKey {
    int curve = 0, device = 0;
    uint64_t hs = 0;                   // hash of n and 64 sampled points: NOMINATES an entry; only a byte-for-byte comparison confirms it
    size_t n = 0;
    reef_msm_ctx *master = nullptr;    // owns the reference on the pre-shifted tables the threads' contexts share
    void *host_copy = nullptr;         // the bytes the resident key was built from
    size_t charged = 0, host_charged = 0;   // device / host bytes charged to the process-wide budgets
    std::atomic<uint64_t> last_use{0};
    std::atomic<int> state{0};         // 0 nominated (seen, no copy), 1 being built, 2 resident, 3 evicted, 4 not worth another try
    uint32_t seen = 1;                 // under the table lock
    ~SharedKey();
};
std::atomic<size_t> g_cache_bytes{0}, g_host_bytes{0};
std::atomic<uint64_t> g_cache_builds{0}, g_cache_hits{0}, g_cache_clones{0}, g_cache_misspeculated{0}, g_cache_spares{0};
std::atomic<uint64_t> g_cache_epoch{0};            // bumped when the table is emptied: threads drop their attachments at their next call
std::atomic<uint64_t> g_seam_calls{0}, g_seam_nominate_ns{0}, g_seam_enqueue_ns{0}, g_seam_confirm_ns{0}, g_seam_wait_ns{0};
static inline uint64_t now_ns() { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// Runs on whichever thread drops the last reference -- never under the table lock (the table hands evicted entries to its caller).
SharedKey::~SharedKey() {
    free(host_copy);
    g_host_bytes -= host_charged;
    if (g_process_exiting.load()) return;          // the driver reclaims everything
    reef_msm_ctx_destroy(master);                  // the last handle on the tables: they are freed here, and only now is the budget released
    g_cache_bytes -= charged;
}
static size_t env_mb(const char *name, unsigned long long dflt) {
    const char *e = getenv(name);
    return (size_t)(e && *e ? strtoull(e, nullptr, 10) : dflt) << 20;
}
static size_t cache_budget() { static const size_t b = env_mb("REEF_MSM_KEY_CACHE_MB", 16384); return b; }
static size_t host_budget() { static const size_t b = env_mb("REEF_MSM_KEY_HOST_MB", 4096); return b; }
// `bytes` of a budget, or nothing: one fetch_add, rolled back when it overshoots (load-then-add let two builders both pass)
static bool reserve_bytes(std::atomic<size_t> &used, size_t bytes, size_t budget) {
    if (used.fetch_add(bytes) + bytes <= budget) return true;
    used.fetch_sub(bytes);
    return false;
}
constexpr size_t KEY_CACHE_MIN_POINTS = 1024, KEY_TABLE_ENTRIES = 16, BG_STREAM_MIN_POINTS = (size_t)1 << 17;
constexpr size_t CMP_PIECE = (size_t)1 << 20, CMP_PARALLEL_MIN = (size_t)4 << 20;

// ---- the bytes of a large key, in pieces, by the caller and the process's helper threads: compared with the retained copy (dst == NULL)
// or copied into a new one (dst != NULL)
struct CompareJob {
    const char *a = nullptr, *b = nullptr;
    char *dst = nullptr;
    size_t bytes = 0, pieces = 0;
    std::atomic<size_t> next{0}, done{0};
    std::atomic<int> differ{0};
    void work() {
        for (;;) {
            const size_t i = next.fetch_add(1, std::memory_order_relaxed);
            if (i >= pieces) return;                   // nothing of a / b is touched beyond this point: the caller may be gone
            const size_t off = i * CMP_PIECE, len = std::min(CMP_PIECE, bytes - off);
            if (dst) memcpy(dst + off, a + off, len);
            else if (!differ.load(std::memory_order_relaxed) && memcmp(a + off, b + off, len) != 0) differ.store(1, std::memory_order_relaxed);
            done.fetch_add(1, std::memory_order_release);
        }
    }
    bool finished() const { return done.load(std::memory_order_acquire) == pieces; }
};
struct ComparePool {
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<CompareJob>> q;
    size_t nthreads = 0;
    ComparePool() {
        const char *e = getenv("REEF_MSM_CMP_THREADS");
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        nthreads = e && *e ? (size_t)std::min<long>(64, std::max<long>(0, atol(e))) : (size_t)std::min(8u, std::max(2u, hw / 4));
        for (size_t i = 0; i < nthreads; ++i)
            std::thread([this] {
                for (;;) {
                    std::shared_ptr<CompareJob> j;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return !q.empty(); });
                        j = std::move(q.front());
                        q.pop_front();
                    }
                    j->work();
                }
            }).detach();                               // they sleep on the queue for the life of the process; never joined (static destructors run after HIP)
    }
    void help(const std::shared_ptr<CompareJob> &j) {
        const size_t helpers = std::min(nthreads, j->pieces > 1 ? j->pieces - 1 : 0);
        if (!helpers) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < helpers; ++i) q.push_back(j);
        }
        cv.notify_all();
    }
};
static ComparePool &compare_pool() {
    static ComparePool *p = new ComparePool();
    return *p;
}
// a copy of the caller's key bytes that outlives the call (malloc; NULL when the host is out of memory)
static void *copy_key_bytes(const reef_affine *points, size_t bytes) {
    void *copy = malloc(bytes);
    if (!copy) return nullptr;
    if (bytes < CMP_PARALLEL_MIN) { memcpy(copy, points, bytes); return copy; }
    auto job = std::make_shared<CompareJob>();
    job->a = (const char *)points; job->dst = (char *)copy; job->bytes = bytes; job->pieces = (bytes + CMP_PIECE - 1) / CMP_PIECE;
    compare_pool().help(job);
    job->work();
    while (!job->finished()) std::this_thread::yield();
    return copy;
}

struct KeyTable {
    std::mutex mu;
    std::vector<std::shared_ptr<SharedKey>> keys;
    std::atomic<uint64_t> tick{0};
    // every entry leaves the table; the entries are destroyed AFTER the lock has been released (their destructors wait for and
    // free device memory), and a key lives on until the last thread attached to it has let go -- which the epoch asks of every thread
    void clear();
};
static KeyTable &key_table() {
    static KeyTable *t = new KeyTable();            // never destroyed: static destructors run after HIP may be gone
    return *t;
}
static inline uint64_t hmix(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
static uint64_t sampled_hash(const reef_affine *p, size_t n) {
    uint64_t h = hmix(n);
    const size_t step = std::max<size_t>(1, n / 63);
    auto take = [&](size_t i) {
        const uint64_t *w = (const uint64_t *)(p + i);
        for (int k = 0; k < 8; ++k) h = hmix(h ^ w[k]) + i;
    };
    for (size_t i = 0; i < n; i += step) take(i);
    take(n - 1);
    return h;
}
// A context of the resident path with everything its first call would otherwise have to create: its own workspace, already sized by an
// MSM of n_warm points, and the host-mapped landing zone the speculative result goes to.
struct ReadyCtx {
    reef_msm_ctx *ctx = nullptr;
    reef_jacobian *pinned = nullptr;
    int device = -1;
    size_t n_warm = 0;
    std::shared_ptr<SharedKey> key;                    // the key ctx is attached to: it (and its share of the budget) lives while any context -- a thread's or a pooled one -- is on its tables
    void destroy() {                                   // HIP work: on the builder thread, or where nobody is waiting
        reef_msm_ctx_destroy(ctx);
        if (pinned) (void)hipHostFree(pinned);
        ctx = nullptr; pinned = nullptr; device = -1; n_warm = 0;
        key.reset();                                   // after the context: the key's destructor frees what the context was reading
    }
};
static reef_status ready_ctx_make(ReadyCtx *r, const std::shared_ptr<SharedKey> &k, size_t n_warm) {
    reef_msm_ctx *master = k->master;
    const int device = k->device;
    r->device = device;
    r->key = k;
    {
        PoolNoGrowth ng;
        REEF_TRY(reef_msm_ctx_clone(&r->ctx, master));
    }
    if (hipHostMalloc((void **)&r->pinned, 128, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        r->pinned = nullptr;
        set_error("hipHostMalloc failed");
        return REEF_ERR_OOM;
    }
    if (n_warm) {                                      // size the workspace: one MSM of the key's length with all-zero scalars
        void *zeros = calloc(n_warm, sizeof(reef_fe));
        if (!zeros) { set_error("host memory exhausted"); return REEF_ERR_OOM; }
        reef_status st = reef_msm(r->ctx, (const reef_fe *)zeros, n_warm, REEF_HOST, true, r->pinned, REEF_DEVICE);
        if (st == REEF_OK) st = reef_msm_ctx_sync(r->ctx);
        free(zeros);
        REEF_TRY(st);
        r->n_warm = n_warm;
    }
    return REEF_OK;
}

// ONE helper thread per process, started by the first call that has work for it.  Jobs: build a key's resident copy and a spare context for it
// (BUILD), pool a context a thread let go of (RETIRE), destroy the pooled ones of a key that left the table (PURGE).  Callers only ever push and
// take; they never wait for a job.  At process exit the atexit hook below lets the job in flight finish and drops the rest -- no HIP call of
// this thread may overlap the runtime's teardown.
struct Builder {
    enum Kind { BUILD, RETIRE, PURGE };
    struct Job {
        Kind kind = BUILD;
        int device = 0;
        std::shared_ptr<SharedKey> key;
        void *copy = nullptr;
        ReadyCtx retire;
    };
    std::mutex mu;
    std::condition_variable cv, idle_cv;
    std::deque<Job> q;
    bool started = false, busy = false;
    std::vector<ReadyCtx> spares[2];                   // per curve, under mu: ready contexts nobody is using (the builder's fresh ones and the ones threads let go of)
    static constexpr size_t SPARES_MAX = 3;            // per curve; one more and the smallest is destroyed
    std::atomic<uint64_t> jobs_done{0};

    void push(Job &&j) {
        std::lock_guard<std::mutex> lk(mu);
        if (g_process_exiting.load()) { drop(j); return; }
        if (!started) {
            started = true;
            atexit([] { builder_instance().quiesce(); });   // registered after HIP's own: runs before the runtime goes away
            std::thread([this] { run(); }).detach();
        }
        q.push_back(std::move(j));
        cv.notify_one();
    }
    // the spare of (curve, device) if its workspace has served at least n points
    bool take(int curve, int device, size_t n, ReadyCtx *out) {
        std::lock_guard<std::mutex> lk(mu);
        auto &v = spares[curve];
        size_t best = v.size();
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i].device == device && v[i].n_warm >= n && (best == v.size() || v[i].n_warm < v[best].n_warm)) best = i;
        if (best == v.size()) return false;
        *out = v[best];
        v.erase(v.begin() + best);
        return true;
    }
    // A context joins the spares.  Contexts are KEPT rather than destroyed: destroying one is ~30 hipFree calls, each of which waits for the device -- beside a
    // caller that keeps the GPU busy that took 30 ms and stretched the caller's calls from 0.9 to 3.6 ms meanwhile (profiles/r06_stateless_pcie_inclusive.txt,
    // the 2^18 row of the first collections).  Returns the context to destroy when the pool is full: the one with the smallest workspace.
    ReadyCtx give(int curve, const ReadyCtx &r) {
        std::lock_guard<std::mutex> lk(mu);
        // a context on the tables of a key that left the table is not kept: it would keep them, and their share of the budget, alive.  (Under mu: clear() marks the keys
        // BEFORE take_all() takes this lock, an eviction BEFORE its PURGE job is pushed -- a context that slips in here is found by either.)
        if (r.key && r.key->state.load() == 3) return r;
        auto &v = spares[curve];
        v.push_back(r);
        if (v.size() <= SPARES_MAX) return ReadyCtx();
        size_t small = 0;
        for (size_t i = 1; i < v.size(); ++i)
            if (v[i].n_warm < v[small].n_warm) small = i;
        const ReadyCtx victim = v[small];
        v.erase(v.begin() + small);
        return victim;
    }
    // every spare leaves (the table was emptied: their clones hold references on the keys' tables); destroyed by the caller
    std::vector<ReadyCtx> take_all() {
        std::lock_guard<std::mutex> lk(mu);
        std::vector<ReadyCtx> all;
        for (auto &v : spares) { all.insert(all.end(), v.begin(), v.end()); v.clear(); }
        return all;
    }
    void wait_idle() {                                 // tests / diagnostics (reef_key_cache_info never waits)
        std::unique_lock<std::mutex> lk(mu);
        idle_cv.wait(lk, [&] { return q.empty() && !busy; });
    }
    static Builder &builder_instance();

  private:
    static void drop(Job &j) {                         // a job that will never run: host memory only (the process is going away)
        free(j.copy);
        j.copy = nullptr;
        if (j.kind == BUILD && j.key) {
            g_cache_bytes -= j.key->charged; g_host_bytes -= j.key->host_charged;
            j.key->charged = j.key->host_charged = 0;
            int expect = 1;
            j.key->state.compare_exchange_strong(expect, 0);
        }
    }
    void quiesce() {
        g_process_exiting.store(true);
        std::unique_lock<std::mutex> lk(mu);
        for (auto &j : q) drop(j);
        q.clear();
        idle_cv.wait(lk, [&] { return !busy; });
    }
    void run() {
        t_pool_no_growth = true;                       // this thread never creates a stream while one exists: its work is ordered on a stream the callers use (common.h)
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !q.empty(); });
                if (g_process_exiting.load()) { for (auto &x : q) drop(x); q.clear(); continue; }
                j = std::move(q.front());
                q.pop_front();
                busy = true;
            }
            static const bool log_jobs = [] { const char *l = getenv("REEF_MSM_LOG"); return l && atoi(l) >= 2; }();
            const uint64_t tj = now_ns();
            const Kind kind = j.kind;
            const size_t npts = j.key ? j.key->n : 0;
            try {
                execute(j);
            } catch (...) {                            // bad_alloc and friends: the key stays on the plain path
                if (j.kind == BUILD) fail_build(j, 4);
            }
            jobs_done += 1;
            if (log_jobs)
                fprintf(stderr, "libreef_msm: builder job %s (%zu points) took %.3f ms, ended at %.3f ms\n", kind == BUILD ? "BUILD" : kind == RETIRE ? "RETIRE" : "PURGE", npts,
                        (now_ns() - tj) * 1e-6, (now_ns() % 100000000000ull) * 1e-6);
            {
                std::lock_guard<std::mutex> lk(mu);
                busy = false;
            }
            idle_cv.notify_all();
        }
    }
    static void fail_build(Job &j, int state) {
        free(j.copy);
        j.copy = nullptr;
        g_cache_bytes -= j.key->charged; g_host_bytes -= j.key->host_charged;
        j.key->charged = j.key->host_charged = 0;
        int expect = 1;                                // an entry evicted meanwhile (state 3) stays evicted
        j.key->state.compare_exchange_strong(expect, state, std::memory_order_release);
    }
    void execute(Job &j) {
        if (j.kind == RETIRE) {                        // kept as a spare; only a full pool costs a destruction
            ReadyCtx victim = give(j.device, j.retire);
            victim.destroy();
            return;
        }
        if (j.kind == PURGE) {                         // a key left the table: pooled contexts still on its tables would keep them (and their share of the budget) alive
            std::vector<ReadyCtx> gone;
            {
                std::lock_guard<std::mutex> lk(mu);
                for (auto &v : spares)
                    for (size_t i = 0; i < v.size();)
                        if (v[i].key && v[i].key->state.load() == 3) { gone.push_back(v[i]); v.erase(v.begin() + i); }
                        else ++i;
            }
            for (auto &r : gone) r.destroy();
            return;
        }
        DeviceGuard dg(j.device);
        if (!dg.ok) { if (j.kind == BUILD) fail_build(j, 4); return; }
        SharedKey &k = *j.key;
        // a long build gets (once per process and device) a stream of this thread's own: on a caller's stream its kernels would keep that
        // caller's next call waiting for the whole build; short ones are not worth a stream creation (common.h)
        if (k.n >= BG_STREAM_MIN_POINTS) (void)stream_pool().ensure_background_stream(k.device);
        reef_msm_opts o = {};
        o.bucket_groups = 1;
        o.byte_tables = 2;                             // never for a key the caller did not create: 256 KiB per point would dwarf the budget
        o.device = k.device;
        reef_msm_ctx *master = nullptr;
        if (reef_msm_ctx_create(&master, k.curve, (const reef_affine *)j.copy, k.n, REEF_HOST, &o) != REEF_OK) { fail_build(j, 4); return; }
        k.host_copy = j.copy;
        k.master = master;                             // (nobody reads it before state 2)
        j.copy = nullptr;
        ReadyCtx spare;
        if (ready_ctx_make(&spare, j.key, k.n) != REEF_OK) spare.destroy();      // never fatal: the first caller makes its own
        g_cache_builds += 1;
        ReadyCtx old;
        if (spare.ctx) {                               // BEFORE the key is published: whoever sees state 2 finds the spare
            old = give(k.curve, spare);
            g_cache_spares += 1;
            if (old.ctx == spare.ctx) { spare.destroy(); old = ReadyCtx(); }      // (the pool was full of larger ones, any of which serves this key: the new spare itself goes)
        }
        int expect = 1;                                // an entry evicted meanwhile (state 3) stays evicted; its destructor frees what was built
        if (!k.state.compare_exchange_strong(expect, 2, std::memory_order_release) && spare.ctx) {
            bool mine = false;                         // nobody will ask for this key: its spare must not keep the tables alive
            {
                std::lock_guard<std::mutex> lk(mu);
                auto &v = spares[k.curve];
                for (size_t i = 0; i < v.size(); ++i)
                    if (v[i].ctx == spare.ctx) { v.erase(v.begin() + i); mine = true; break; }
            }
            if (mine) spare.destroy();
        }
        if (old.ctx) old.destroy();                    // the pool was full: its smallest context goes
    }
};
Builder &Builder::builder_instance() {
    static Builder *b = new Builder();                 // never destroyed
    return *b;
}
static Builder &builder() { return Builder::builder_instance(); }

void KeyTable::clear() {
    std::vector<std::shared_ptr<SharedKey>> gone;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (auto &k : keys) k->state.store(3);
        gone.swap(keys);
    }
    g_cache_epoch += 1;
    for (auto &r : builder().take_all()) r.destroy();
}

struct TlsCtx {
    reef_msm_ctx *ctx[2] = {nullptr, nullptr};      // plain path: re-keyed on every call
    ReadyCtx r[2];                                  // resident path: this thread's stream, workspace and landing zone, attached to whichever shared key a call nominates
    int ctx_dev[2] = {-1, -1};
    uint64_t epoch = 0;
    void drop_resident() {
        for (int c = 0; c < 2; ++c) r[c].destroy();
    }
    ~TlsCtx() {
        if (g_process_exiting.load()) return;          // process teardown: never call into HIP (the keys' destructors look at the same flag)
        for (auto *c : ctx) reef_msm_ctx_destroy(c);
        for (int c = 0; c < 2; ++c) {                  // a worker thread that ends hands its resident contexts to the pool: the next worker takes them as they are
            if (r[c].ctx) {
                Builder::Job j;
                j.kind = Builder::RETIRE; j.retire = r[c]; j.device = c;
                r[c] = ReadyCtx();
                builder().push(std::move(j));
            }
        }
    }
};
thread_local TlsCtx g_tls;

static reef_status pippenger_plain(int curve, reef_jacobian *out, const reef_affine *points, size_t npoints, const reef_fe *scalars, bool is_mont) {
    reef_msm_ctx *&c = g_tls.ctx[curve];
    int dev = 0;
    REEF_HIP_TRY(hipGetDevice(&dev));
    if (c && g_tls.ctx_dev[curve] != dev) { reef_msm_ctx_destroy(c); c = nullptr; }   // the caller moved to another GPU
    if (!c) {
        PoolNoGrowth ng;                               // the context takes a stream when it has work; a second stream only when two callers really overlap
        REEF_TRY(reef_msm_ctx_create(&c, curve, points, npoints, REEF_HOST, nullptr));
        g_tls.ctx_dev[curve] = dev;
    } else {
        REEF_TRY(vt(curve)->ctx_rekey(c->impl, points, npoints, REEF_HOST));
    }
    return reef_msm(c, scalars, npoints, REEF_HOST, is_mont, out, REEF_HOST);
}

// The call that brought the key's second appearance (state 1 is ours) has been served: reserve the budgets, keep the bytes, hand the
// build to the builder thread.  Failure is never fatal -- the key keeps being served on the plain path.
static void schedule_build(const std::shared_ptr<SharedKey> &k, const reef_affine *points) {
    const size_t bytes = k->n * sizeof(reef_affine);
    uint32_t T = 1;
    (void)reef_msm_plan_for(k->n, 0, 1, nullptr, nullptr, nullptr, &T);
    const size_t cost = bytes * (size_t)T;             // T pre-shifted tables
    int back_to = 0;                                   // a full budget: nominated again, tried again when the key returns
    if (reserve_bytes(g_cache_bytes, cost, cache_budget())) {
        if (reserve_bytes(g_host_bytes, bytes, host_budget())) {
            void *copy = copy_key_bytes(points, bytes);
            if (copy) {
                k->charged = cost;
                k->host_charged = bytes;
                Builder::Job j;
                j.kind = Builder::BUILD; j.device = k->device; j.key = k; j.copy = copy;
                builder().push(std::move(j));
                return;
            }
            back_to = 4;                               // the host is out of memory: not worth another try
            g_host_bytes -= bytes;
        }
        g_cache_bytes -= cost;
    }
    int expect = 1;                                    // an entry evicted meanwhile (state 3) stays evicted
    k->state.compare_exchange_strong(expect, back_to, std::memory_order_release);
}

// the caller's bytes against a key's retained copy, without an MSM beside it (choosing among several resident keys that share their samples)
static bool bytes_equal(const reef_affine *points, const SharedKey &k, size_t bytes) {
    if (bytes < CMP_PARALLEL_MIN) return memcmp(points, k.host_copy, bytes) == 0;
    auto job = std::make_shared<CompareJob>();
    job->a = (const char *)points; job->b = (const char *)k.host_copy; job->bytes = bytes; job->pieces = (bytes + CMP_PIECE - 1) / CMP_PIECE;
    compare_pool().help(job);
    job->work();
    while (!job->finished()) std::this_thread::yield();
    return job->differ.load() == 0;
}

// The call on the resident key `k`, speculatively: *same = the caller's bytes are the key's (then *out holds the result).
static reef_status pippenger_resident(const std::shared_ptr<SharedKey> &k, reef_jacobian *out, const reef_affine *points, size_t npoints,
                                      const reef_fe *scalars, bool is_mont, bool *same) {
    const size_t bytes = npoints * sizeof(reef_affine);
    const int curve = k->curve;
    ReadyCtx &r = g_tls.r[curve];
    auto retire = [&] {                                // destroyed by the builder thread, not here
        Builder::Job j;
        j.kind = Builder::RETIRE; j.retire = r; j.device = curve;      // (for a RETIRE job `device` carries the curve: the pool is per curve)
        r = ReadyCtx();                                // (the retired context keeps its key -- tables and budget -- until it is attached elsewhere or destroyed)
        builder().push(std::move(j));
    };
    if (r.ctx && r.device != k->device) retire();
    if (!r.ctx || r.n_warm < npoints) {                // the spare the builder prepared for a key of this size
        ReadyCtx s;
        if (builder().take(curve, k->device, npoints, &s)) {
            if (r.ctx) retire();
            r = s;
        }
    }
    if (!r.ctx) {                                      // no spare (another thread took it): this thread makes its own
        const reef_status st = ready_ctx_make(&r, k, 0);
        if (st != REEF_OK) { r.destroy(); return st; }
        g_cache_clones += 1;
    }
    if (r.key != k) REEF_TRY(reef_msm_ctx_attach(r.ctx, k->master));   // O(1): the thread's stream and workspace on another key's tables
    r.key = k;                                         // the key this context let go of may end here (its last reference): outside every lock
    reef_msm_ctx *c = r.ctx;
    const uint64_t t0 = now_ns();
    std::shared_ptr<CompareJob> job;
    if (bytes >= CMP_PARALLEL_MIN) {                   // a large key: the helpers start on it while this thread stages the scalars
        job = std::make_shared<CompareJob>();
        job->a = (const char *)points; job->b = (const char *)k->host_copy; job->bytes = bytes; job->pieces = (bytes + CMP_PIECE - 1) / CMP_PIECE;
        compare_pool().help(job);
    }
    const reef_status issued = reef_msm(c, scalars, npoints, REEF_HOST, is_mont, r.pinned, REEF_DEVICE);   // enqueued; the result goes to host-mapped memory
    const uint64_t t1 = now_ns();
    bool eq;
    if (job) {                                         // the caller's pointers must outlive every piece in flight, whatever `issued` says
        job->work();
        while (!job->finished()) std::this_thread::yield();
        eq = job->differ.load() == 0;
    } else {
        eq = memcmp(points, k->host_copy, bytes) == 0; // while the GPU works
    }
    const uint64_t t2 = now_ns();
    REEF_TRY(issued);
    REEF_TRY(reef_msm_ctx_sync(c));
    const uint64_t t3 = now_ns();
    r.n_warm = std::max(r.n_warm, npoints);
    g_seam_calls += 1; g_seam_enqueue_ns += t1 - t0; g_seam_confirm_ns += t2 - t1; g_seam_wait_ns += t3 - t2;
    *same = eq;
    if (eq) {
        memcpy(out, r.pinned, sizeof(reef_jacobian));
        g_cache_hits += 1;
    } else {
        g_cache_misspeculated += 1;
    }
    return REEF_OK;
}

static reef_status pippenger_cached(int curve, reef_jacobian *out, const reef_affine *points, size_t npoints, const reef_fe *scalars, bool is_mont) {
    int dev = 0;
    REEF_HIP_TRY(hipGetDevice(&dev));
    KeyTable &tab = key_table();
    {
        const uint64_t ep = g_cache_epoch.load();
        if (g_tls.epoch != ep) {                       // the table was emptied (reef_key_cache_clear, or another thread's allocation failure)
            g_tls.drop_resident();
            g_tls.epoch = ep;
        }
    }
    const uint64_t tn = now_ns();
    const uint64_t hs = sampled_hash(points, npoints);
    // Since round 6 the samples are the ONLY hash: rounds 3-5 also hashed every byte of a key that was not (yet) resident -- 0.06 ms per MiB on every such
    // call, 5 of the 11 ms of a 2^20-point one, and on EVERY call for bases that never return (the IPA's folded generators through the zero-patch route).
    // Nothing rested on it: a resident key is only ever used after its retained bytes have been compared with the caller's, and a key that is nominated
    // for a resident copy by its samples alone is built from the bytes of the call that nominates it.
    std::vector<std::shared_ptr<SharedKey>> tried;     // resident keys the samples nominated whose bytes turned out to be another key's
    std::shared_ptr<SharedKey> k;
    bool builder_of = false;
    std::shared_ptr<SharedKey> evicted;                // destroyed after the table lock has been released (declared before it is taken)
    bool first_pass = true;
    for (;;) {
        // resident keys the samples nominate and this call has not yet compared itself with, most recently used first.  A LOOP, because the table changes while
        // a call is outside the lock: a key that another thread's call handed to the builder may be published between this call's look at the table and its turn
        // to register a (second) appearance -- it must then be compared with, not nominated a second time (two threads that both mis-speculated on a key's twin
        // while the key itself was being published built it twice: tests/test_gpu_concurrent.py, 3 runs in 8).
        std::vector<std::shared_ptr<SharedKey>> cands;
        {
            std::lock_guard<std::mutex> lk(tab.mu);
            for (auto &e : tab.keys)
                if (e->curve == curve && e->device == dev && e->n == npoints && e->hs == hs && e->state.load(std::memory_order_acquire) == 2 &&
                    std::find(tried.begin(), tried.end(), e) == tried.end())
                    cands.push_back(e);
            std::sort(cands.begin(), cands.end(), [](const std::shared_ptr<SharedKey> &x, const std::shared_ptr<SharedKey> &y) { return x->last_use.load() > y->last_use.load(); });
            if (!cands.empty()) {
                cands[0]->last_use.store(++tab.tick);
            } else {
                // nothing (left) to compare with: register this appearance -- under the SAME lock that found no untried resident key
                const uint64_t now = ++tab.tick;
                for (auto &e : tab.keys)               // an entry the samples nominate that has no resident copy (yet)
                    if (e->curve == curve && e->device == dev && e->n == npoints && e->hs == hs && e->state.load() != 2) k = e;
                if (k) {
                    k->last_use.store(now);
                    k->seen += 1;
                    int expect = 0;
                    builder_of = k->seen >= 2 && k->state.compare_exchange_strong(expect, 1);   // second appearance: worth a resident copy (built from THIS call's bytes)
                } else {
                    if (tab.keys.size() >= KEY_TABLE_ENTRIES) {  // forget the least recently used key; keys seen once (no resident copy) go first
                        size_t lru = tab.keys.size();
                        for (int pass = 0; pass < 2 && lru == tab.keys.size(); ++pass)
                            for (size_t i = 0; i < tab.keys.size(); ++i) {
                                const int st = tab.keys[i]->state.load();
                                if (st == 1 || (pass == 0 && st == 2)) continue;              // never the one being built
                                if (lru == tab.keys.size() || tab.keys[i]->last_use.load() < tab.keys[lru]->last_use.load()) lru = i;
                            }
                        if (lru < tab.keys.size()) {
                            tab.keys[lru]->state.store(3);
                            evicted = std::move(tab.keys[lru]);
                            tab.keys.erase(tab.keys.begin() + lru);
                        }
                    }
                    if (tab.keys.size() < KEY_TABLE_ENTRIES) {
                        auto e = std::make_shared<SharedKey>();
                        e->curve = curve; e->device = dev; e->n = npoints; e->hs = hs;
                        e->last_use.store(now);
                        tab.keys.push_back(e);
                    }
                }
            }
        }
        if (first_pass) g_seam_nominate_ns += now_ns() - tn;
        if (cands.empty()) break;
        if (cands.size() == 1 && first_pass) {             // the usual case: speculate -- the MSM runs while the bytes are compared
            bool same = false;
            REEF_TRY(pippenger_resident(cands[0], out, points, npoints, scalars, is_mont, &same));
            if (same) return REEF_OK;
        } else {                                           // several resident keys agree on all 64 samples (or a late one appeared): the bytes choose before an MSM is spent
            for (auto &c : cands)
                if (bytes_equal(points, *c, npoints * sizeof(reef_affine))) {
                    {
                        std::lock_guard<std::mutex> lk(tab.mu);
                        c->last_use.store(++tab.tick);
                    }
                    bool same = false;
                    REEF_TRY(pippenger_resident(c, out, points, npoints, scalars, is_mont, &same));
                    if (same) return REEF_OK;
                    break;
                }
        }
        tried.insert(tried.end(), cands.begin(), cands.end());
        first_pass = false;
    }
    if (evicted) {                                     // pooled contexts still attached to it are destroyed by the builder thread
        Builder::Job j;
        j.kind = Builder::PURGE;
        builder().push(std::move(j));
    }
    evicted.reset();                                   // here: HIP work of the destructor (if this was the last reference) outside the lock
    static const bool log_calls = [] { const char *l = getenv("REEF_MSM_LOG"); return l && atoi(l) >= 2; }();
    const uint64_t tp0 = log_calls ? now_ns() : 0;
    const reef_status st = pippenger_plain(curve, out, points, npoints, scalars, is_mont);   // also while the builder is at work on this key
    const uint64_t tp1 = log_calls ? now_ns() : 0;
    if (builder_of) {
        if (st == REEF_OK) schedule_build(k, points);
        else { int expect = 1; k->state.compare_exchange_strong(expect, 0); }
    }
    if (log_calls)
        fprintf(stderr, "libreef_msm: plain-path call (%zu points): nominate %.3f ms, plain MSM %.3f ms, hand-over to the builder %.3f ms\n", npoints, (tp0 - tn) * 1e-6,
                (tp1 - tp0) * 1e-6, (now_ns() - tp1) * 1e-6);
    return st;
}

static reef_status pippenger_try(int curve, reef_jacobian *out, const reef_affine *points, size_t npoints, const reef_fe *scalars, bool is_mont) {
    static const bool cache_on = !(getenv("REEF_MSM_KEY_CACHE") && atoi(getenv("REEF_MSM_KEY_CACHE")) == 0);
    hook_process_exit();
    reef_status st = (!cache_on || npoints < KEY_CACHE_MIN_POINTS) ? pippenger_plain(curve, out, points, npoints, scalars, is_mont)
                                                                     : pippenger_cached(curve, out, points, npoints, scalars, is_mont);
    if (st == REEF_ERR_OOM) {                           // give the cache's memory back and serve the call uncached
        builder().wait_idle();                          // a build in flight holds memory too; its key is gone from the table after clear()
        key_table().clear();                            // every thread lets go of its attachment at its next call (epoch)
        g_tls.drop_resident();
        g_tls.epoch = g_cache_epoch.load();
        st = pippenger_plain(curve, out, points, npoints, scalars, is_mont);
    }
    return st;
}

static void pippenger(int curve, reef_jacobian *out, const reef_affine *points, size_t npoints, const reef_fe *scalars, bool is_mont) {
    if (pippenger_try(curve, out, points, npoints, scalars, is_mont) != REEF_OK) {
        fprintf(stderr, "libreef_msm: mult_pippenger_%s failed: %s\n", curve == REEF_PALLAS ? "pallas" : "vesta", reef_last_error());
        abort();
    }
}
}  // namespace

void reef_key_cache_info(reef_key_cache_stats *out) {
    if (!out) return;
    KeyTable &tab = key_table();
    std::lock_guard<std::mutex> lk(tab.mu);
    out->entries = tab.keys.size();
    out->resident_keys = 0;
    for (auto &k : tab.keys) out->resident_keys += k->state.load() == 2;
    out->resident_bytes = g_cache_bytes.load();
    out->builds = g_cache_builds.load();
    out->hits = g_cache_hits.load();
    out->clones = g_cache_clones.load();
    out->misspeculated = g_cache_misspeculated.load();
    out->spares = g_cache_spares.load();
}
void reef_key_cache_clear(void) { key_table().clear(); }
void reef_key_cache_wait(void) { builder().wait_idle(); }
void reef_key_cache_timing_get(reef_key_cache_timing *out, int reset) {
    if (out) {
        memset(out, 0, sizeof *out);
        out->calls = g_seam_calls.load(); out->nominate_ns = g_seam_nominate_ns.load(); out->enqueue_ns = g_seam_enqueue_ns.load();
        out->confirm_ns = g_seam_confirm_ns.load(); out->wait_ns = g_seam_wait_ns.load();
    }
    if (reset) { g_seam_calls = 0; g_seam_nominate_ns = 0; g_seam_enqueue_ns = 0; g_seam_confirm_ns = 0; g_seam_wait_ns = 0; }
}

void mult_pippenger_pallas(reef_jacobian *out, const reef_affine *points, size_t npoints, const reef_fe *scalars, bool is_mont) {
    pippenger(REEF_PALLAS, out, points, npoints, scalars, is_mont);
}
void mult_pippenger_vesta(reef_jacobian *out, const reef_affine *points, size_t npoints, const reef_fe *scalars, bool is_mont) {
    pippenger(REEF_VESTA, out, points, npoints, scalars, is_mont);
}

}  // extern "C"
