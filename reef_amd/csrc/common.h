// Host-side plumbing shared by the per-curve engines and the C ABI (not part of the ABI).
#pragma once
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/reef_msm.h"

namespace reef {

void set_error(const char *fmt, ...) __attribute__((format(printf, 1, 2)));

#define REEF_HIP_TRY(expr)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            ::reef::set_error("%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);  \
            return e_ == hipErrorOutOfMemory ? REEF_ERR_OOM : REEF_ERR_HIP;                         \
        }                                                                                           \
    } while (0)

#define REEF_TRY(expr)                          \
    do {                                        \
        reef_status s_ = (expr);                \
        if (s_ != REEF_OK) return s_;           \
    } while (0)

// Environment switches come in two kinds.  SUPPORTED ones are policy an embedder may want to set and are documented in INTEGRATION.md
// ("Environment"): REEF_MSM_KEY_CACHE, REEF_MSM_KEY_CACHE_MB, REEF_MSM_KEY_HOST_MB, REEF_MSM_CMP_THREADS, REEF_MSM_STREAMS, REEF_MSM_HW_QUEUES,
// REEF_MSM_LOG, REEF_MSM_WIDE, REEF_MSM_WIDE_MAX_LOG, REEF_MSM_HOST_COMBINE, REEF_MSM_GRAPH, REEF_MSM_WARM, REEF_SC_FENCE, REEF_RCCL_LIB -- read with getenv.  EXPERIMENTAL ones exist
// for A/B measurements (tools/, profiles/) and for tests that force a code path at sizes the oracle can handle; they are read through
// exp_env and exist only in builds with -DREEF_EXPERIMENT: libreef_msm_exp.so (csrc/Makefile), which a few tests and tools/ load on request.
// libreef_msm.so -- what ships, what the GPU suite tests and bench.py measures -- is built without it: there an undocumented variable in the
// environment changes nothing.
inline const char *exp_env(const char *name) {
#ifdef REEF_EXPERIMENT
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// Allocations and launches of a context must target the device its key lives on, whatever the calling thread's current device is.  A
// hipSetDevice that FAILS must not go unnoticed -- the work that follows would silently go to the caller's device: entry points
// open the scope with REEF_ON_DEVICE and return REEF_ERR_HIP.
struct DeviceGuard {
    int prev = -1;
    bool switched = false, ok = true;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
        ok = err == hipSuccess;
        if (!ok) (void)hipGetLastError();
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define REEF_ON_DEVICE(dev)                                                                                   \
    ::reef::DeviceGuard dg_((dev));                                                                           \
    if (!dg_.ok) {                                                                                            \
        ::reef::set_error("cannot make device %d current: %s", (int)(dev), hipGetErrorString(dg_.err));       \
        return REEF_ERR_HIP;                                                                                  \
    }

// Grow-only device buffer (steady state performs no allocation).
// Bumped whenever a workspace buffer is (re)allocated or released: captured hipGraphs hold raw device pointers and are
// dropped when the generation they were captured under is gone (engine.inc, run_core_graphed).
inline std::atomic<uint64_t> g_devbuf_gen{0};
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    reef_status ensure(size_t bytes) {
        if (bytes <= cap) return REEF_OK;
        ++g_devbuf_gen;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes + bytes / 8 + 256;
        REEF_HIP_TRY(hipMalloc(&p, want));
        cap = want;
        return REEF_OK;
    }
    void release() {
        if (p) { (void)hipFree(p); ++g_devbuf_gen; }
        p = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// ---- the library's HIP streams: one small process-wide pool ------------------------------------------------------------
// The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues by COUNT: a new stream goes to the queue with
// the fewest streams, whether those are busy or idle, and keeps it for life; streams on one queue run in turn.  A stream per
// context therefore lets idle contexts push the busy ones of a long-lived process onto shared queues -- measured in round 4:
// the final SNARK's three arguments, issued from three threads at once, took 9.6 ms inside bench.py's process where the bare
// harness takes 6.1 ms, and 6.2 ms in the same process once an earlier replay had shifted the runtime's counts
// (tools/diag_queues2.py, profiles/r04_concurrency_bisect.txt).  So contexts do not own streams any more: the library keeps at
// most REEF_MSM_STREAMS (default 8) streams per device, created one at a time when every existing one has a caller at work, and
// a context (or a stateless call) takes the stream with the fewest callers AT WORK each time it starts from idle -- all its
// earlier work has been waited for -- and keeps it until it is idle again.  Sharing a stream only adds ordering, never removes it.
// hipStreamCreate is the most expensive call the library makes after start-up: 7-9 ms per stream while the runtime still has hardware queues to
// set up (the first GPU_MAX_HW_QUEUES streams of the process), 2 ms afterwards -- and while it runs, every HIP call of every OTHER thread stalls
// up to 0.8 ms (reef_amd/csrc/tools/stream_probe.hip, profiles/r06_stream_probe.txt: a 20-launch MSM beside two creations read 15 ms).  So the
// pool grows only for callers that are really concurrent.  A thread that sets this flag (the drop-in symbols' builder thread, for life; the
// symbols' own per-thread contexts while they are created: api.cpp) shares the least loaded existing stream instead of creating one.
inline thread_local bool t_pool_no_growth = false;
struct PoolNoGrowth {
    bool prev;
    PoolNoGrowth() : prev(t_pool_no_growth) { t_pool_no_growth = true; }
    ~PoolNoGrowth() { t_pool_no_growth = prev; }
};
struct PoolStream {
    hipStream_t s = nullptr;
    int device = 0;
    std::atomic<int> active{0};      // callers between their first enqueue and the wait that found them idle again
    std::atomic<int> pins{0};        // contexts whose stream was handed to the caller (reef_msm_ctx_stream): they stay here for life
    bool bg_only = false;            // created for the no-growth threads alone (ensure_background_stream): callers never take it
    std::atomic<int> background{0};  // no-growth threads at work here (the drop-in symbols' builder): they share a stream, and a caller that finds only
                                     // THEM on it shares it too instead of creating the next one (a creation on the caller's thread: 8 ms, measured)
};
struct StreamPool {
    std::mutex mu;
    std::vector<PoolStream *> streams;               // entries are never moved or destroyed: the pointers handed out stay valid
    static size_t limit() {
        static const size_t v = [] { const char *e = getenv("REEF_MSM_STREAMS"); const long n = e ? atol(e) : 8; return (size_t)(n < 1 ? 1 : n > 64 ? 64 : n); }();
        return v;
    }
    // A stream is created OUTSIDE the pool lock (hipStreamCreate takes milliseconds while the runtime still has hardware queues to set up, and
    // since round 6 a helper thread creates contexts beside the callers: api.cpp, Builder): the callers' pick() never waits behind one.
    static PoolStream *make(int device, hipError_t *err) {
        PoolStream *p = new PoolStream();
        p->device = device;
        *err = hipStreamCreateWithFlags(&p->s, hipStreamNonBlocking);   // the caller's current device is `device` (DeviceGuard)
        if (*err != hipSuccess) { delete p; (void)hipGetLastError(); return nullptr; }
        static const bool log = [] { const char *l = getenv("REEF_MSM_LOG"); return l && atoi(l) >= 2; }();
        if (log) fprintf(stderr, "libreef_msm: stream created on device %d\n", device);
        return p;
    }
    // a stream reserved for a graph capture (engine.inc, run_core_graphed) is nobody else's until the capture has ended
    static constexpr int RESERVED = 1 << 20;
    static int load_of(const PoolStream *p, bool for_pin) {
        const int a = 2 * p->active.load(std::memory_order_relaxed) + p->background.load(std::memory_order_relaxed);
        return for_pin ? 2000 * p->pins.load(std::memory_order_relaxed) + a : a;
    }
    // the stream of `device` with the smallest load (callers at work; for_pin: contexts pinned to it first), a new one while every stream has
    // work and the pool may grow.  A stream under capture is skipped; when every stream is (REEF_MSM_STREAMS=1 ...) the caller waits for the
    // capture to end -- it lasts microseconds.
    reef_status pick_impl(int device, PoolStream **out, bool for_pin) {
        for (;;) {
            bool grow = false;
            {
                std::lock_guard<std::mutex> lk(mu);
                PoolStream *best = nullptr;
                size_t on_device = 0;
                for (PoolStream *p : streams)
                    if (p->device == device) {
                        if (p->bg_only) {                      // the builder's own stream, once it exists: its work goes there and nowhere else
                            if (t_pool_no_growth && !for_pin) { best = p; break; }
                            continue;
                        }
                        ++on_device;
                        if (p->active.load(std::memory_order_relaxed) >= RESERVED) continue;
                        if (!best || load_of(p, for_pin) < load_of(best, for_pin)) best = p;
                    }
                grow = !t_pool_no_growth && on_device < limit() && (!best || (for_pin ? load_of(best, true) > 0 : best->active.load(std::memory_order_relaxed) > 0));
                if (!grow && best) {
                    count_in(best, for_pin);
                    *out = best;
                    return REEF_OK;
                }
                if (!grow && on_device == 0) grow = true;     // limit() >= 1: cannot happen, but never spin on an empty pool
            }
            if (!grow) { std::this_thread::yield(); continue; }   // every stream is under capture
            hipError_t e = hipSuccess;
            PoolStream *p = make(device, &e);
            std::lock_guard<std::mutex> lk(mu);
            if (p) {
                count_in(p, for_pin);
                streams.push_back(p);
                *out = p;
                return REEF_OK;
            }
            PoolStream *best = nullptr;                       // no new stream to be had: share the least loaded one
            for (PoolStream *q : streams)
                if (q->device == device && !q->bg_only && q->active.load(std::memory_order_relaxed) < RESERVED && (!best || load_of(q, for_pin) < load_of(best, for_pin))) best = q;
            if (!best) { set_error("hipStreamCreate: %s", hipGetErrorString(e)); return REEF_ERR_HIP; }
            count_in(best, for_pin);
            *out = best;
            return REEF_OK;
        }
    }
    static void count_in(PoolStream *p, bool for_pin) {
        if (for_pin) p->pins.fetch_add(1, std::memory_order_relaxed);
        else if (t_pool_no_growth) p->background.fetch_add(1, std::memory_order_relaxed);
        else p->active.fetch_add(1, std::memory_order_relaxed);
    }
    // counted for the caller from here on (*background: as a no-growth thread): leave() with the same flag when idle again
    reef_status pick(int device, PoolStream **out, bool *background = nullptr) {
        if (background) *background = t_pool_no_growth;
        return pick_impl(device, out, false);
    }
    // a stream for a context that will STAY on it (its stream is handed to the caller, e.g. for RCCL ordering): the one with the
    // fewest such contexts, then the fewest callers at work -- three pinned contexts of a bench rank must not share one stream
    reef_status pick_for_pin(int device, PoolStream **out) { return pick_impl(device, out, true); }
    // Graph capture needs the stream to itself: reserve it iff the caller (counted once in `active`, pinned or not) is its only user -- one
    // compare-and-swap, so no other caller can slip in between the check and the reservation.
    bool try_reserve(PoolStream *p, int own_pins) {
        std::lock_guard<std::mutex> lk(mu);                // picks count themselves in under this lock: none can land between the checks
        if (p->pins.load(std::memory_order_relaxed) > own_pins || p->background.load(std::memory_order_relaxed) > 0) return false;
        int expect = 1;
        return p->active.compare_exchange_strong(expect, 1 + RESERVED, std::memory_order_acq_rel);
    }
    static void unreserve(PoolStream *p) { p->active.fetch_sub(RESERVED, std::memory_order_acq_rel); }
    static void leave(PoolStream *p, bool background = false) {
        if (!p) return;
        if (background) p->background.fetch_sub(1, std::memory_order_relaxed);
        else p->active.fetch_sub(1, std::memory_order_relaxed);
    }
    // Creating a stream is expensive while the runtime still has hardware queues to create (several ms each, measured: the first
    // concurrent use of three contexts read 23 ms instead of 6), so it is done where contexts are CREATED, never in a hot call if
    // it can be helped: a new context makes sure the pool holds as many streams as there are contexts alive (up to the limit).
    std::atomic<int> contexts_alive{0};
    reef_status context_created(int device) {
        const size_t want = std::min<size_t>((size_t)std::max(1, ++contexts_alive), limit());
        if (t_pool_no_growth) return REEF_OK;              // an internal context of the drop-in symbols: it takes a stream when it has work (pick)
        return ensure_streams(device, want);
    }
    // A stream of `device` that only no-growth threads use, created (by such a thread, outside the lock) the first time its work is long
    // enough to be worth ~8 ms of creation: ordered on a caller's stream, the table build of a 2^20-point key kept the caller's next call
    // waiting 20 ms (profiles/r06_seam_first_calls.txt).  Beyond REEF_MSM_STREAMS: it is not the callers'.
    reef_status ensure_background_stream(int device) {
        {
            std::lock_guard<std::mutex> lk(mu);
            for (PoolStream *p : streams)
                if (p->device == device && p->bg_only) return REEF_OK;
        }
        // at the LOWEST priority the device offers: the builder's kernels yield to the callers' wherever the hardware lets them
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);              // lo = the numerically largest = the lowest priority
        PoolStream *p = new PoolStream();
        p->device = device;
        hipError_t e = hipStreamCreateWithPriority(&p->s, hipStreamNonBlocking, lo);
        if (e != hipSuccess) { delete p; (void)hipGetLastError(); set_error("hipStreamCreate: %s", hipGetErrorString(e)); return REEF_ERR_HIP; }
        p->bg_only = true;
        std::lock_guard<std::mutex> lk(mu);
        streams.push_back(p);
        return REEF_OK;
    }
    // at least `want` streams on `device` (each created outside the lock)
    reef_status ensure_streams(int device, size_t want) {
        want = std::min(want, limit());
        for (;;) {
            {
                std::lock_guard<std::mutex> lk(mu);
                size_t on_device = 0;
                for (PoolStream *p : streams) on_device += p->device == device && !p->bg_only;
                if (on_device >= want) return REEF_OK;
            }
            hipError_t e = hipSuccess;
            PoolStream *p = make(device, &e);
            if (!p) { set_error("hipStreamCreate: %s", hipGetErrorString(e)); return REEF_ERR_HIP; }
            std::lock_guard<std::mutex> lk(mu);
            streams.push_back(p);                          // two threads may both add one: a stream more than wanted, never fewer
        }
    }
    void context_destroyed() { --contexts_alive; }
};
inline StreamPool &stream_pool() {
    static StreamPool *p = new StreamPool();            // never destroyed: static destructors run after HIP may be gone
    return *p;
}
// What a context (MSM or sum-check) holds of the pool.  Protocol, under the context's own lock: enter() before the first enqueue
// of a call, then idle() once the stream has been waited for (hipStreamSynchronize) -- a call that returns with work in flight
// simply does not call it and the context stays on its stream.  pin(): the stream was handed to the caller (reef_msm_ctx_stream,
// to order RCCL / torch work after it) and must not change any more.
struct StreamLease {
    PoolStream *ps = nullptr;
    bool counted = false, pinned = false, bg = false;
    reef_status enter(int device, hipStream_t *stream) {
        if (!counted) {
            if (pinned && ps) {                              // its own stream for life -- but not while another context captures a graph on it
                for (int a = ps->active.load(std::memory_order_relaxed);;) {
                    if (a >= StreamPool::RESERVED) { std::this_thread::yield(); a = ps->active.load(std::memory_order_relaxed); continue; }
                    if (ps->active.compare_exchange_weak(a, a + 1, std::memory_order_acq_rel)) break;
                }
                bg = false;
            }
            else REEF_TRY(stream_pool().pick(device, &ps, &bg));
            counted = true;
        }
        *stream = ps->s;
        return REEF_OK;
    }
    void idle() {
        if (counted) StreamPool::leave(ps, bg);
        counted = false;
    }
    // from now on the context stays on one stream; only from idle (the caller of reef_msm_ctx_stream has waited for its work)
    reef_status pin(int device, hipStream_t *stream) {
        if (!pinned) {
            if (counted) { *stream = ps->s; pinned = true; ps->pins.fetch_add(1, std::memory_order_relaxed); return REEF_OK; }   // work in flight: this stream it is
            REEF_TRY(stream_pool().pick_for_pin(device, &ps));
            pinned = true;
        }
        *stream = ps->s;
        return REEF_OK;
    }
    void unpin() {
        if (pinned && ps) ps->pins.fetch_sub(1, std::memory_order_relaxed);
        pinned = false;
    }
};

// Per-curve entry points, implemented once per curve in kernels_<curve>.hip via engine.inc.
// One block of a Merkle tree built by several devices: the subtree over the document symbols [index_base, index_base + n) carried up exactly
// `levels` levels above its bottom level (a ragged last block keeps hashing (node, 0) where the whole tree would: merkle_tree.rs:82-114), or --
// nodes_in != NULL -- the levels above a given bottom level (the blocks' roots).  level_out[h], h = 0..levels: where level h of the block goes
// (HOST memory, the caller's form; entries may be NULL); top_out: the first node of the last level (host).
struct MerkleSlice {
    uint64_t index_base = 0;
    uint32_t levels = 0;
    const reef_fe *nodes_in = nullptr;   // host, nodes_n of them, in the caller's form
    size_t nodes_n = 0;
    reef_fe *const *level_out = nullptr;
    reef_fe *top_out = nullptr;
};

struct CurveVTable {
    reef_status (*ctx_create)(void **impl, const reef_affine *bases, size_t n, int loc, const reef_msm_opts *opts);
    reef_status (*ctx_rekey)(void *impl, const reef_affine *bases, size_t n, int loc);
    reef_status (*ctx_clone)(void **impl, void *src);
    reef_status (*ctx_attach)(void *impl, void *src_impl);
    void (*ctx_destroy)(void *impl);
    reef_status (*ctx_sync)(void *impl);
    void *(*ctx_stream)(void *impl);
    reef_status (*ctx_timing)(void *impl, float *total_ms, float *acc_ms);
    reef_status (*ctx_enable_timing)(void *impl, int on);
    reef_status (*ctx_window_split)(void *impl, uint32_t rank, uint32_t world);
    reef_status (*ctx_timing_stats)(void *impl, int reset, uint64_t *calls, double *total_ms, double *acc_ms);
    reef_status (*ctx_sum_points)(void *impl, const reef_jacobian *in, size_t n, reef_jacobian *out);
    reef_status (*ctx_plan)(void *impl, uint32_t *c, uint32_t *w, uint32_t *g, uint32_t *t);
    int (*ctx_byte_tables)(void *impl);
    reef_status (*msm)(void *impl, const reef_fe *scalars, size_t n, int loc, bool is_mont, reef_jacobian *out, int out_loc);
    reef_status (*msm_rows)(void *impl, const reef_fe *scalars, size_t rows, size_t row_len, int loc, bool is_mont,
                            uint32_t max_bits, const reef_fe *blinds, const reef_affine *h, reef_jacobian *out, int out_loc);
    reef_status (*msm_rows_symbols)(void *impl, const uint8_t *symbols, size_t rows, size_t row_len, int loc, uint32_t bits,
                                    const reef_fe *blinds, const reef_affine *h, bool blinds_are_mont, reef_jacobian *out, int out_loc);
    reef_status (*ipa_cross)(void *impl, const reef_fe *a, size_t n_k, int loc, bool is_mont, const reef_fe *w1s, const reef_fe *w2s, size_t k,
                             reef_jacobian *out_l, reef_jacobian *out_r);
    reef_status (*msm_folded)(void *impl, const reef_fe *v, size_t len, size_t off, int loc, bool is_mont, const reef_fe *w1s, const reef_fe *w2s,
                              size_t k, reef_jacobian *out, int out_loc);
    reef_status (*fold)(const reef_affine *gens, size_t half, int loc, const reef_fe *w1, const reef_fe *w2, reef_affine *out);
    reef_status (*normalize)(const reef_jacobian *in, size_t n, int loc, reef_affine *out_aff, uint8_t *out_comp);
    reef_status (*sum_points)(const reef_jacobian *in, size_t n, int loc, reef_jacobian *out);
    reef_status (*gen_bases)(uint64_t k0, uint64_t d, size_t n, reef_affine *out, int loc);
    reef_status (*gen_scalars)(uint64_t seed, int kind, uint64_t small_bound, size_t n, bool to_mont, reef_fe *out, int loc);
    reef_status (*test_field_op)(int op, const reef_fe *a, const reef_fe *b, reef_fe *out, size_t n);  // coordinate field
    reef_status (*test_ec_op)(int op, const reef_affine *p, const reef_affine *q, const reef_fe *k, reef_jacobian *out, size_t n);
    reef_status (*bench_fmul)(uint32_t iters, double *per_s);
    // row N2: sum-check vector kernels over the curve's scalar field
    reef_status (*sc_create)(void **impl, size_t len);
    void (*sc_destroy)(void *impl);
    reef_status (*sc_set)(void *impl, int which, const reef_fe *vals, size_t n, int loc);
    reef_status (*sc_read)(void *impl, int which, size_t count, reef_fe *out_host);
    reef_status (*sc_coeffs)(void *impl, size_t pow, reef_fe *out3_host);
    reef_status (*sc_fold)(void *impl, size_t pow, const reef_fe *r);
    reef_status (*sc_fold_coeffs)(void *impl, size_t pow, const reef_fe *r, reef_fe *out3_host);
    reef_status (*sc_gen_eq)(void *impl, const reef_fe *rs, const uint32_t *qs, size_t nq, const reef_fe *last_q, size_t ell);
    reef_status (*sc_reset)(void *impl);
    reef_status (*sc_sync)(void *impl);
    // row N3: bound rows / evaluation of a multilinear table over the curve's scalar field
    reef_status (*mle_bound)(const void *z, size_t n, int elem_bytes, int z_loc, bool is_mont, const reef_fe *point, size_t num_vars,
                             size_t left_vars, reef_fe *lz_out, int out_loc, reef_fe *eval_out);
    // row N4: Poseidon Merkle commitment over the curve's scalar field.  slice (may be NULL: the whole tree of a document): one block of a tree that
    // several devices build together (reef_merkle_commit_devices, api.cpp)
    reef_status (*merkle_commit)(const reef_poseidon_params *pp, const uint32_t *doc, size_t n, int doc_loc, bool is_mont, reef_fe *tree_out,
                                 int out_loc, reef_fe *root_out, const struct MerkleSlice *slice);
    // row N1: commitment-key derivation (hash to the curve; coordinates in the curve's base field)
    reef_status (*derive_generators)(const uint8_t *label, size_t label_len, size_t n, const reef_keygen_params *kp, bool is_mont, reef_affine *out,
                                     int out_loc);
    reef_status (*plan_for)(size_t n, uint32_t c_opt, uint32_t g_opt, uint32_t *c, uint32_t *w, uint32_t *g, uint32_t *t);
};

const CurveVTable *pallas_vtable();
const CurveVTable *vesta_vtable();

}  // namespace reef
