// Device groups (include/reef_msm.h section 5): one MSM split by Pippenger window or by points, or the rows of a Hyrax
// commitment dealt out whole, over several GPUs of ONE process.  Composition only: every member is a reef_msm_ctx of the handle
// API, this file adds the host threads that issue the members' calls side by side and the exchange of the 96-byte partial sums
// (peer copies between devices, or host-mapped slots as the labelled fallback).  No arithmetic here.
//
// Why one process: Reef's prover is a single Rust process (src/backend/main.rs:82, src/backend/framework.rs:81-166) and the only
// way it can reach eight GPUs is through calls it makes itself; reef_amd/distributed.py (one process per GPU over
// torch.distributed) stays the harness the driver's torchrun launches, this is what a Rust binding calls.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <rccl/rccl.h>   // declarations only: the library is opened at run time (Rccl below), nothing links against it

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <chrono>
#include "common.h"

using namespace reef;

namespace {

// device scopes: common.h (DeviceGuard / REEF_ON_DEVICE -- a hipSetDevice that fails is an error, never a silent wrong device)

// REEF_EXCHANGE_RCCL: librccl opened at run time.  One table per process; a failed load is remembered with its message.
struct Rccl {
    void *handle = nullptr;
    char why[384] = "";
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    bool ok() const { return handle != nullptr; }
};
static Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *name = getenv("REEF_RCCL_LIB");
        if (!name || !*name) name = "librccl.so.1";
        void *h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
        if (!h) {
            const char *e = dlerror();
            snprintf(r.why, sizeof r.why, "dlopen(%s): %s", name, e ? e : "failed");
            return;
        }
        bool all = true;
        auto sym = [&](const char *s) -> void * {
            void *p = dlsym(h, s);
            if (!p && all) { snprintf(r.why, sizeof r.why, "%s has no symbol %s", name, s); all = false; }
            return p;
        };
        r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.Send = (decltype(r.Send))sym("ncclSend");
        r.Recv = (decltype(r.Recv))sym("ncclRecv");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        r.GetVersion = (decltype(r.GetVersion))sym("ncclGetVersion");
        if (!all) { dlclose(h); return; }
        r.handle = h;
    });
    return r;
}
#define REEF_RCCL_TRY(call)                                                                           \
    do {                                                                                              \
        const ncclResult_t r_ = (call);                                                               \
        if (r_ != ncclSuccess) { set_error("%s: %s", #call, rccl().GetErrorString(r_)); return REEF_ERR_HIP; } \
    } while (0)

// One persistent host thread per member: a call on the group hands every member its share and returns when all have ENQUEUED
// (or, for rows, finished) theirs.  hipMemcpyAsync from the caller's pageable memory blocks the issuing thread while the runtime
// stages it, so one issuing thread would feed eight PCIe links one after the other.
struct Workers {
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    std::function<void(size_t)> job;
    uint64_t generation = 0;
    size_t pending = 0;
    bool stop = false;
    explicit Workers(size_t n) {
        for (size_t i = 0; i < n; ++i) threads.emplace_back([this, i] { loop(i); });
    }
    ~Workers() {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv_go.notify_all();
        for (auto &t : threads) t.join();
    }
    void loop(size_t i) {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv_go.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            auto f = job;
            lk.unlock();
            f(i);
            lk.lock();
            if (--pending == 0) cv_done.notify_all();
        }
    }
    // f(i) on worker i for every i; the caller's thread waits
    void run(const std::function<void(size_t)> &f) {
        std::unique_lock<std::mutex> lk(mu);
        job = f;
        pending = threads.size();
        ++generation;
        cv_go.notify_all();
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};

struct Member {
    int device = 0;
    reef_msm_ctx *ctx = nullptr;        // the member's share of a split MSM: window split (i, ndev) or its slice of the points
    reef_msm_ctx *whole = nullptr;      // windows groups: a clone without the split, for rows that are dealt out whole
    hipStream_t stream = nullptr;       // ctx's stream (pinned to it: the exchange is ordered on it)
    hipEvent_t done = nullptr;          // recorded after the member's partial sum has been sent
    reef_jacobian *partial = nullptr;   // members on another device than member 0: where the partial sum is computed before it is sent
    void *stage = nullptr;              // device-resident inputs of the caller live on devices[0]: this member's copy
    size_t stage_cap = 0;
    size_t off = 0, len = 0;            // points split: the slice [off, off + len) of the key
    int rank = 0;                       // RCCL: the member's device as a rank of the group's communicator (member 0's device is rank 0)
    hipEvent_t t_begin = nullptr, t_end = nullptr;   // timing mode: around the member's share on its stream
    double issue_ms = 0;                // timing mode: host time of the member's issue call (staging of pageable scalars, peer fetches, launches)
};

}  // namespace

struct reef_msm_group {
    int curve = 0;
    uint32_t split = 0, exchange = 0, distinct = 0, peer_members = 0;
    size_t n = 0;
    std::vector<Member> m;
    reef_jacobian *gather = nullptr;    // PEER: ndev slots on member 0's device; HOST: ndev slots of host-mapped pinned memory
    reef_jacobian *landing = nullptr;   // host-mapped: the sum lands here
    Workers *workers = nullptr;
    std::vector<ncclComm_t> comms;      // RCCL: one communicator handle per distinct device, in order of first appearance in devices[]
    std::vector<size_t> rank_lead;      // RCCL: the first member on each rank's device (its stream carries the rank's sends)
    std::mutex mu;                      // a group serialises its calls
    // round 6: the first run on several devices cannot be rehearsed, so a call can say where its time went (reef_msm_group_enable_timing)
    bool timing = false;
    reef_msm_group_timing last = {};
    std::chrono::steady_clock::time_point t_call;
    // round 6: window split from HOST scalars, fan-out variant (reef_msm_group_opts.scalars = REEF_SCALARS_FANOUT): one upload to devices[0], peer copies from there
    uint32_t scalars_mode = 0;
    void *fan = nullptr;                // devices[0]: the uploaded scalars
    size_t fan_cap = 0;
    hipEvent_t fan_ready = nullptr;     // recorded on member 0's stream after the upload
    std::atomic<bool> fan_posted{false}; // this call's upload has been enqueued and its event recorded (members spin the microseconds until then)
};

namespace {

// The members' calls fail on their own threads: the message is carried back to the caller's thread-local slot.
struct Outcome {
    std::mutex mu;
    reef_status st = REEF_OK;
    char msg[512] = "";
    void note(reef_status s) {
        if (s == REEF_OK) return;
        std::lock_guard<std::mutex> lk(mu);
        if (st != REEF_OK) return;
        st = s;
        snprintf(msg, sizeof msg, "%s", reef_last_error());
    }
    reef_status finish() {
        if (st != REEF_OK) set_error("group member: %s", msg);
        return st;
    }
};

static reef_status member_stage(Member &mb, size_t bytes) {
    if (bytes <= mb.stage_cap) return REEF_OK;
    if (mb.stage) { (void)hipFree(mb.stage); mb.stage = nullptr; mb.stage_cap = 0; }
    REEF_HIP_TRY(hipMalloc(&mb.stage, bytes + bytes / 8 + 256));
    mb.stage_cap = bytes + bytes / 8 + 256;
    return REEF_OK;
}
// src (device memory of devices[0]) -> a pointer the member's kernels can read, ordered on the member's stream
static reef_status member_fetch(reef_msm_group *g, Member &mb, const void *src, size_t bytes, const void **out) {
    if (mb.device == g->m[0].device || bytes == 0) { *out = src; return REEF_OK; }
    REEF_TRY(member_stage(mb, bytes));
    REEF_HIP_TRY(hipMemcpyPeerAsync(mb.stage, mb.device, src, g->m[0].device, bytes, mb.stream));
    *out = mb.stage;
    return REEF_OK;
}

// Where member i's partial sum goes, and -- after the call that computes it has been enqueued -- how it reaches member 0.
static reef_jacobian *partial_target(reef_msm_group *g, size_t i) {
    Member &mb = g->m[i];
    if (g->exchange == REEF_EXCHANGE_RCCL) return i == 0 ? g->gather : mb.partial;      // every other member sends, also to its own device
    if (g->exchange == REEF_EXCHANGE_HOST || mb.device == g->m[0].device) return g->gather + i;
    return mb.partial;
}
static reef_status partial_send(reef_msm_group *g, size_t i) {
    Member &mb = g->m[i];
    if (g->exchange == REEF_EXCHANGE_PEER && mb.device != g->m[0].device)
        REEF_HIP_TRY(hipMemcpyPeerAsync(g->gather + i, g->m[0].device, mb.partial, mb.device, sizeof(reef_jacobian), mb.stream));
    if (g->exchange == REEF_EXCHANGE_PEER && i > 0) REEF_HIP_TRY(hipEventRecord(mb.done, mb.stream));
    // RCCL: the sends of a rank ride on its leading member's stream (one stream per communicator inside a group call); the other
    // members of the device hand over with an event
    if (g->exchange == REEF_EXCHANGE_RCCL && i != g->rank_lead[(size_t)mb.rank]) REEF_HIP_TRY(hipEventRecord(mb.done, mb.stream));
    return REEF_OK;
}
// RCCL: all members have enqueued.  One group call: member i's 96 bytes go from its rank to rank 0's slot i.
static reef_status rccl_gather(reef_msm_group *g) {
    Rccl &R = rccl();
    const size_t nd = g->m.size();
    for (size_t i = 1; i < nd; ++i) {
        Member &mb = g->m[i];
        const size_t lead = g->rank_lead[(size_t)mb.rank];
        if (lead == i) continue;
        REEF_ON_DEVICE(mb.device);
        REEF_HIP_TRY(hipStreamWaitEvent(g->m[lead].stream, mb.done, 0));
    }
    REEF_RCCL_TRY(R.GroupStart());
    reef_status st = REEF_OK;
    for (size_t i = 1; i < nd && st == REEF_OK; ++i) {
        Member &mb = g->m[i];
        ncclResult_t r;
        {
            DeviceGuard dg(mb.device);
            if (!dg.ok) { set_error("cannot make device %d current: %s", mb.device, hipGetErrorString(dg.err)); st = REEF_ERR_HIP; break; }
            r = R.Send(mb.partial, sizeof(reef_jacobian), ncclUint8, 0, g->comms[(size_t)mb.rank], g->m[g->rank_lead[(size_t)mb.rank]].stream);
        }
        if (r == ncclSuccess) {
            DeviceGuard dg(g->m[0].device);
            if (!dg.ok) { set_error("cannot make device %d current: %s", g->m[0].device, hipGetErrorString(dg.err)); st = REEF_ERR_HIP; break; }
            r = R.Recv(g->gather + i, sizeof(reef_jacobian), ncclUint8, mb.rank, g->comms[0], g->m[0].stream);
        }
        if (r != ncclSuccess) { set_error("ncclSend/ncclRecv of member %zu: %s", i, R.GetErrorString(r)); st = REEF_ERR_HIP; }
    }
    const ncclResult_t e = R.GroupEnd();                 // closed whatever happened inside, or the thread stays in group mode
    if (st == REEF_OK && e != ncclSuccess) { set_error("ncclGroupEnd: %s", R.GetErrorString(e)); st = REEF_ERR_HIP; }
    return st;
}
// All members have enqueued: the partial sums are added on member 0's device and the result is waited for.
static double ms_between(std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }
static reef_status combine_untimed(reef_msm_group *g, reef_jacobian *out);
static reef_status combine(reef_msm_group *g, reef_jacobian *out) {
    if (!g->timing) return combine_untimed(g, out);
    // timing mode: every member is waited for on its own (its events say what its stream spent), then the usual combine
    reef_status st = REEF_OK;
    for (auto &mb : g->m) {
        DeviceGuard dg(mb.device);
        if (hipEventSynchronize(mb.t_end) != hipSuccess) { (void)hipGetLastError(); set_error("hipEventSynchronize failed on member device %d", mb.device); st = REEF_ERR_HIP; }
    }
    g->last.members_done_ms = ms_between(g->t_call, std::chrono::steady_clock::now());
    for (size_t i = 0; i < g->m.size() && i < 16; ++i) {
        Member &mb = g->m[i];
        DeviceGuard dg(mb.device);
        float ms = 0;
        if (st == REEF_OK && hipEventElapsedTime(&ms, mb.t_begin, mb.t_end) != hipSuccess) { (void)hipGetLastError(); ms = -1; }
        g->last.member_stream_ms[i] = ms;
        g->last.member_issue_ms[i] = mb.issue_ms;
    }
    const auto tc = std::chrono::steady_clock::now();
    if (st == REEF_OK) st = combine_untimed(g, out);
    const auto te = std::chrono::steady_clock::now();
    g->last.combine_ms = ms_between(tc, te);
    g->last.total_ms = ms_between(g->t_call, te);
    return st;
}
static reef_status combine_untimed(reef_msm_group *g, reef_jacobian *out) {
    const size_t nd = g->m.size();
    Member &m0 = g->m[0];
    REEF_ON_DEVICE(m0.device);
    if (g->exchange == REEF_EXCHANGE_HOST) {
        reef_status st = REEF_OK;
        for (auto &mb : g->m) {                        // the slots are host memory: every member's last kernel must have finished
            const reef_status w = reef_msm_ctx_sync(mb.ctx);
            if (st == REEF_OK) st = w;
        }
        REEF_TRY(st);
    } else if (g->exchange == REEF_EXCHANGE_RCCL) {
        if (nd > 1) REEF_TRY(rccl_gather(g));
    } else {
        for (size_t i = 1; i < nd; ++i) REEF_HIP_TRY(hipStreamWaitEvent(m0.stream, g->m[i].done, 0));
    }
    if (nd == 1) {
        REEF_TRY(reef_msm_ctx_sync(m0.ctx));
        if (g->exchange == REEF_EXCHANGE_HOST) { memcpy(out, g->gather, sizeof *out); return REEF_OK; }
        return reef_memcpy(out, g->gather, sizeof *out, REEF_HOST, REEF_DEVICE);
    }
    REEF_TRY(reef_msm_ctx_sum_points(m0.ctx, g->gather, nd, g->landing));
    REEF_TRY(reef_msm_ctx_sync(m0.ctx));
    memcpy(out, g->landing, sizeof *out);
    return REEF_OK;
}

static void group_free(reef_msm_group *g) {
    if (!g) return;
    delete g->workers;                                 // joins the member threads first
    for (auto &mb : g->m)
        if (mb.ctx) { DeviceGuard dg(mb.device); (void)reef_msm_ctx_sync(mb.ctx); }
    for (ncclComm_t c : g->comms)
        if (c) (void)rccl().CommDestroy(c);
    for (auto &mb : g->m) {
        DeviceGuard dg(mb.device);
        if (mb.ctx) (void)reef_msm_ctx_sync(mb.ctx);
        reef_msm_ctx_destroy(mb.whole);
        reef_msm_ctx_destroy(mb.ctx);
        if (mb.done) (void)hipEventDestroy(mb.done);
        if (mb.t_begin) (void)hipEventDestroy(mb.t_begin);
        if (mb.t_end) (void)hipEventDestroy(mb.t_end);
        if (mb.partial) (void)hipFree(mb.partial);
        if (mb.stage) (void)hipFree(mb.stage);
    }
    if (!g->m.empty()) {
        DeviceGuard dg(g->m[0].device);
        if (g->fan) (void)hipFree(g->fan);
        if (g->fan_ready) (void)hipEventDestroy(g->fan_ready);
        if (g->gather) {
            if (g->exchange == REEF_EXCHANGE_HOST) (void)hipHostFree(g->gather);
            else (void)hipFree(g->gather);
        }
    }
    if (g->landing) (void)hipHostFree(g->landing);
    delete g;
}

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

}  // namespace

extern "C" {

reef_status reef_msm_group_create(reef_msm_group **out, int curve, const reef_affine *bases, size_t n, int bases_loc, const reef_msm_opts *key_opts,
                                  const int *devices, size_t ndev, const reef_msm_group_opts *gopts) {
    if (!out || !devices || (n && !bases)) { set_error("reef_msm_group_create: null argument"); return REEF_ERR_ARG; }
    if (curve != REEF_PALLAS && curve != REEF_VESTA) { set_error("unknown curve %d", curve); return REEF_ERR_ARG; }
    if (ndev == 0 || ndev > 64) { set_error("reef_msm_group_create: %zu members (1..64)", ndev); return REEF_ERR_ARG; }
    const uint32_t split = gopts ? gopts->split : (uint32_t)REEF_SPLIT_WINDOWS;
    uint32_t exchange = gopts ? gopts->exchange : (uint32_t)REEF_EXCHANGE_DEFAULT;
    if (split != REEF_SPLIT_WINDOWS && split != REEF_SPLIT_POINTS) { set_error("reef_msm_group_create: unknown split %u", split); return REEF_ERR_ARG; }
    if (exchange > REEF_EXCHANGE_RCCL) { set_error("reef_msm_group_create: unknown exchange %u", exchange); return REEF_ERR_ARG; }
    if (exchange == REEF_EXCHANGE_DEFAULT) exchange = REEF_EXCHANGE_PEER;
    const uint32_t scalars_mode = gopts ? gopts->scalars : (uint32_t)REEF_SCALARS_EACH;
    if (scalars_mode > REEF_SCALARS_FANOUT) { set_error("reef_msm_group_create: unknown scalars mode %u", scalars_mode); return REEF_ERR_ARG; }
    for (size_t i = 0; i < ndev; ++i)
        if (devices[i] < 0) { set_error("reef_msm_group_create: devices[%zu] = %d", i, devices[i]); return REEF_ERR_ARG; }
    const int visible = reef_device_count();
    if (visible <= 0) { set_error("no HIP device visible"); return REEF_ERR_NO_GPU; }
    for (size_t i = 0; i < ndev; ++i)
        if (devices[i] >= visible) { set_error("reef_msm_group_create: devices[%zu] = %d, %d visible", i, devices[i], visible); return REEF_ERR_ARG; }
    if (exchange == REEF_EXCHANGE_RCCL && !rccl().ok()) {
        set_error("reef_msm_group_create: REEF_EXCHANGE_RCCL asked for and RCCL cannot be loaded: %s", rccl().why);
        return REEF_ERR_HIP;
    }
    return guarded([&]() -> reef_status {
        reef_msm_group *g = new reef_msm_group();
        g->curve = curve; g->split = split; g->exchange = exchange; g->n = n; g->scalars_mode = scalars_mode;
        g->m.resize(ndev);
        reef_status st = [&]() -> reef_status {
            const int dev0 = devices[0];
            for (size_t i = 0; i < ndev; ++i) {
                Member &mb = g->m[i];
                mb.device = devices[i];
                bool first_on_device = true;
                for (size_t j = 0; j < i; ++j) first_on_device = first_on_device && devices[j] != devices[i];
                if (first_on_device) g->rank_lead.push_back(i);
                for (size_t r = 0; r < g->rank_lead.size(); ++r)
                    if (devices[g->rank_lead[r]] == devices[i]) mb.rank = (int)r;
                g->distinct += first_on_device;
                REEF_ON_DEVICE(mb.device);
                reef_msm_opts o = {};
                if (key_opts) o = *key_opts;
                o.device = mb.device;
                if (split == REEF_SPLIT_WINDOWS) {
                    // the whole key on every DEVICE, once: later members of a device are clones of its first member
                    size_t first = i;
                    for (size_t j = 0; j < i; ++j)
                        if (devices[j] == devices[i]) { first = j; break; }
                    if (first != i) {
                        REEF_TRY(reef_msm_ctx_clone(&mb.ctx, g->m[first].whole));
                    } else if (bases_loc == REEF_DEVICE && mb.device != dev0 && n) {
                        void *tmp = nullptr;
                        REEF_HIP_TRY(hipMalloc(&tmp, n * sizeof(reef_affine)));
                        hipError_t e = hipMemcpyPeer(tmp, mb.device, bases, dev0, n * sizeof(reef_affine));
                        reef_status s = REEF_OK;
                        if (e != hipSuccess) { set_error("hipMemcpyPeer: %s", hipGetErrorString(e)); s = REEF_ERR_HIP; }
                        if (s == REEF_OK) s = reef_msm_ctx_create(&mb.ctx, curve, (const reef_affine *)tmp, n, REEF_DEVICE, &o);
                        (void)hipFree(tmp);
                        REEF_TRY(s);
                    } else {
                        REEF_TRY(reef_msm_ctx_create(&mb.ctx, curve, bases, n, bases_loc, &o));
                    }
                    REEF_TRY(reef_msm_ctx_clone(&mb.whole, mb.ctx));
                    // the rows context stays on ONE stream of the pool too: the first pageable copy a stream carries pays for the runtime's staging
                    // buffers (5-20 ms, once per stream), and a member that wandered over the pool paid it again on every new stream it met
                    if (!reef_msm_ctx_stream(mb.whole)) { set_error("reef_msm_group_create: member %zu got no stream for its rows", i); return REEF_ERR_HIP; }
                    REEF_TRY(reef_msm_ctx_set_window_split(mb.ctx, (uint32_t)i, (uint32_t)ndev));
                    mb.off = 0; mb.len = n;
                } else {
                    mb.off = n * i / ndev;
                    mb.len = n * (i + 1) / ndev - mb.off;
                    const reef_affine *src = bases ? bases + mb.off : nullptr;
                    if (bases_loc == REEF_DEVICE && mb.device != dev0 && mb.len) {
                        void *tmp = nullptr;
                        REEF_HIP_TRY(hipMalloc(&tmp, mb.len * sizeof(reef_affine)));
                        hipError_t e = hipMemcpyPeer(tmp, mb.device, src, dev0, mb.len * sizeof(reef_affine));
                        reef_status s = REEF_OK;
                        if (e != hipSuccess) { set_error("hipMemcpyPeer: %s", hipGetErrorString(e)); s = REEF_ERR_HIP; }
                        if (s == REEF_OK) s = reef_msm_ctx_create(&mb.ctx, curve, (const reef_affine *)tmp, mb.len, REEF_DEVICE, &o);
                        (void)hipFree(tmp);
                        REEF_TRY(s);
                    } else {
                        static const reef_affine none = {};
                        REEF_TRY(reef_msm_ctx_create(&mb.ctx, curve, mb.len ? src : &none, mb.len, mb.len ? bases_loc : REEF_HOST, &o));
                    }
                }
                mb.stream = (hipStream_t)reef_msm_ctx_stream(mb.ctx);
                if (!mb.stream) { set_error("reef_msm_group_create: member %zu got no stream", i); return REEF_ERR_HIP; }
                REEF_HIP_TRY(hipEventCreateWithFlags(&mb.done, hipEventDisableTiming));
                if (mb.device != dev0 || (exchange == REEF_EXCHANGE_RCCL && i > 0)) REEF_HIP_TRY(hipMalloc((void **)&mb.partial, sizeof(reef_jacobian)));
                if (mb.device != dev0) {
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, mb.device, dev0) == hipSuccess && can) {
                        const hipError_t e = hipDeviceEnablePeerAccess(dev0, 0);
                        if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) g->peer_members += 1;
                        (void)hipGetLastError();
                    } else {
                        (void)hipGetLastError();
                    }
                }
            }
            {
                REEF_ON_DEVICE(dev0);
                if (exchange == REEF_EXCHANGE_HOST) REEF_HIP_TRY(hipHostMalloc((void **)&g->gather, ndev * sizeof(reef_jacobian), hipHostMallocPortable | hipHostMallocMapped));
                else REEF_HIP_TRY(hipMalloc((void **)&g->gather, ndev * sizeof(reef_jacobian)));
                REEF_HIP_TRY(hipHostMalloc((void **)&g->landing, sizeof(reef_jacobian), hipHostMallocPortable | hipHostMallocMapped));
            }
            if (exchange == REEF_EXCHANGE_RCCL) {
                std::vector<int> devlist;
                for (size_t lead : g->rank_lead) devlist.push_back(devices[lead]);
                g->comms.assign(devlist.size(), nullptr);
                REEF_RCCL_TRY(rccl().CommInitAll(g->comms.data(), (int)devlist.size(), devlist.data()));
            }
            if (ndev > 1) g->workers = new Workers(ndev);
            return REEF_OK;
        }();
        if (st != REEF_OK) {
            char keep[512];
            snprintf(keep, sizeof keep, "%s", reef_last_error());
            group_free(g);
            set_error("%s", keep);
            return st;
        }
        *out = g;
        return REEF_OK;
    });
}

void reef_msm_group_destroy(reef_msm_group *grp) {
    if (!grp) return;
    { std::lock_guard<std::mutex> lk(grp->mu); }       // a call in flight on another thread finishes first
    group_free(grp);
}

reef_status reef_msm_group_enable_timing(reef_msm_group *grp, int on) {
    if (!grp) { set_error("null argument"); return REEF_ERR_ARG; }
    std::lock_guard<std::mutex> lk(grp->mu);
    if (on)
        for (auto &mb : grp->m) {
            if (mb.t_begin) continue;
            REEF_ON_DEVICE(mb.device);
            REEF_HIP_TRY(hipEventCreate(&mb.t_begin));
            REEF_HIP_TRY(hipEventCreate(&mb.t_end));
        }
    grp->timing = on != 0;
    return REEF_OK;
}
reef_status reef_msm_group_last_timing(reef_msm_group *grp, reef_msm_group_timing *out) {
    if (!grp || !out) { set_error("null argument"); return REEF_ERR_ARG; }
    std::lock_guard<std::mutex> lk(grp->mu);
    if (!grp->timing || grp->last.members == 0) { set_error("no timed group call yet (reef_msm_group_enable_timing first)"); return REEF_ERR_ARG; }
    *out = grp->last;
    return REEF_OK;
}

reef_status reef_msm_group_info_get(reef_msm_group *grp, reef_msm_group_info *info) {
    if (!grp || !info) { set_error("null argument"); return REEF_ERR_ARG; }
    memset(info, 0, sizeof *info);
    info->members = (uint32_t)grp->m.size();
    info->distinct_devices = grp->distinct;
    info->split = grp->split;
    info->exchange = grp->exchange;
    info->peer_members = grp->peer_members;
    for (size_t i = 0; i < grp->m.size() && i < 16; ++i) {
        bool shared = false;                           // windows groups: the members of one device share one resident copy
        if (grp->split == REEF_SPLIT_WINDOWS)
            for (size_t j = 0; j < i; ++j) shared = shared || grp->m[j].device == grp->m[i].device;
        info->key_points[i] = shared ? 0 : grp->m[i].len;
    }
    return REEF_OK;
}

}  // extern "C"

// One call split over the members: issue(i, target) enqueues member i's share with its partial sum going to `target`.
static reef_status split_call(reef_msm_group *g, const std::function<reef_status(size_t, reef_jacobian *)> &issue, reef_jacobian *out) {
    Outcome oc;
    const bool timing = g->timing;
    if (timing) {
        memset(&g->last, 0, sizeof g->last);
        g->last.members = (uint32_t)g->m.size();
        g->t_call = std::chrono::steady_clock::now();
    }
    auto one = [&](size_t i) {
        Member &mb = g->m[i];
        DeviceGuard dg(mb.device);
        reef_status st = REEF_OK;
        if (!dg.ok) { set_error("cannot make device %d current: %s", mb.device, hipGetErrorString(dg.err)); st = REEF_ERR_HIP; }
        const auto t0 = std::chrono::steady_clock::now();
        if (timing && st == REEF_OK && hipEventRecord(mb.t_begin, mb.stream) != hipSuccess) { set_error("hipEventRecord failed"); st = REEF_ERR_HIP; }
        if (st == REEF_OK) st = issue(i, partial_target(g, i));
        if (st == REEF_OK) st = partial_send(g, i);
        if (timing && st == REEF_OK && hipEventRecord(mb.t_end, mb.stream) != hipSuccess) { set_error("hipEventRecord failed"); st = REEF_ERR_HIP; }
        if (timing) mb.issue_ms = ms_between(t0, std::chrono::steady_clock::now());
        oc.note(st);
    };
    if (g->workers) g->workers->run(one);
    else one(0);
    if (timing) g->last.distribute_ms = ms_between(g->t_call, std::chrono::steady_clock::now());
    if (oc.st != REEF_OK) {                            // whatever was enqueued is waited for before the buffers can be reused
        for (auto &mb : g->m) (void)reef_msm_ctx_sync(mb.ctx);
        return oc.finish();
    }
    return combine(g, out);
}

extern "C" reef_status reef_msm_group_msm(reef_msm_group *grp, const reef_fe *scalars, size_t n, int scalars_loc, bool is_mont, reef_jacobian *out) {
    if (!grp || !out || (n && !scalars)) { set_error("null argument"); return REEF_ERR_ARG; }
    if (n > grp->n) { set_error("n = %zu exceeds the key length %zu", n, grp->n); return REEF_ERR_ARG; }
    std::lock_guard<std::mutex> lk(grp->mu);
    return guarded([&]() -> reef_status {
        if (scalars_loc == REEF_HOST && grp->split == REEF_SPLIT_WINDOWS && grp->scalars_mode == REEF_SCALARS_FANOUT && grp->m.size() > 1 && n) {
            // ONE upload to devices[0] (on member 0's stream), an event, and every other member's peer copy waits for it on its own stream
            Member &m0 = grp->m[0];
            const size_t bytes = n * sizeof(reef_fe);
            {
                REEF_ON_DEVICE(m0.device);
                if (bytes > grp->fan_cap) {
                    if (grp->fan) { (void)hipFree(grp->fan); grp->fan = nullptr; grp->fan_cap = 0; }
                    REEF_HIP_TRY(hipMalloc(&grp->fan, bytes + bytes / 8 + 256));
                    grp->fan_cap = bytes + bytes / 8 + 256;
                }
                if (!grp->fan_ready) REEF_HIP_TRY(hipEventCreateWithFlags(&grp->fan_ready, hipEventDisableTiming));
            }
            const void *fan = grp->fan;
            grp->fan_posted.store(false, std::memory_order_release);
            return split_call(grp, [&, fan, bytes](size_t i, reef_jacobian *target) -> reef_status {
                Member &mb = grp->m[i];
                const void *src = fan;
                if (i == 0) {
                    struct Post {                      // whatever happens to the upload, the other members must not wait for ever (a failed call's result is discarded)
                        std::atomic<bool> &f;
                        ~Post() { f.store(true, std::memory_order_release); }
                    } post{grp->fan_posted};
                    REEF_HIP_TRY(hipMemcpyAsync(grp->fan, scalars, bytes, hipMemcpyHostToDevice, mb.stream));
                    REEF_HIP_TRY(hipEventRecord(grp->fan_ready, mb.stream));
                    grp->fan_posted.store(true, std::memory_order_release);
                } else {
                    while (!grp->fan_posted.load(std::memory_order_acquire)) std::this_thread::yield();   // the event must have been RECORDED before it is waited for
                    REEF_HIP_TRY(hipStreamWaitEvent(mb.stream, grp->fan_ready, 0));
                    REEF_TRY(member_stage(mb, bytes));
                    REEF_HIP_TRY(hipMemcpyPeerAsync(mb.stage, mb.device, fan, grp->m[0].device, bytes, mb.stream));   // also for a member on devices[0]: the calls a node runs
                    src = mb.stage;
                }
                return reef_msm(mb.ctx, (const reef_fe *)src, n, REEF_DEVICE, is_mont, target, REEF_DEVICE);
            }, out);
        }
        return split_call(grp, [&](size_t i, reef_jacobian *target) -> reef_status {
            Member &mb = grp->m[i];
            size_t off = 0, cnt = n;                   // windows: every member takes all the scalars
            if (grp->split == REEF_SPLIT_POINTS) {
                off = std::min(mb.off, n);
                cnt = std::min(mb.off + mb.len, n) - off;
            }
            const void *src = scalars ? scalars + off : nullptr;
            if (scalars_loc == REEF_DEVICE) REEF_TRY(member_fetch(grp, mb, src, cnt * sizeof(reef_fe), &src));
            return reef_msm(mb.ctx, (const reef_fe *)src, cnt, scalars_loc, is_mont, target, REEF_DEVICE);
        }, out);
    });
}

// rows > 1: member i computes rows [r0, r1) whole (its clone without the window split) and writes them into the caller's array.
template <class Call> static reef_status dealt_rows(reef_msm_group *g, size_t rows, const Call &call) {
    Outcome oc;
    const size_t nd = g->m.size();
    auto one = [&](size_t i) {
        const size_t r0 = rows * i / nd, r1 = rows * (i + 1) / nd;
        if (r1 == r0) return;
        DeviceGuard dg(g->m[i].device);
        if (!dg.ok) { set_error("cannot make device %d current: %s", g->m[i].device, hipGetErrorString(dg.err)); oc.note(REEF_ERR_HIP); return; }
        oc.note(call(i, r0, r1 - r0));
    };
    if (g->workers) g->workers->run(one);
    else one(0);
    return oc.finish();
}

extern "C" {

reef_status reef_msm_group_rows(reef_msm_group *grp, const reef_fe *scalars, size_t rows, size_t row_len, int scalars_loc, bool is_mont,
                                uint32_t max_scalar_bits, const reef_fe *blinds, const reef_affine *h, reef_jacobian *out) {
    if (!grp || !out || ((rows * row_len) && !scalars) || (blinds && !h)) { set_error("null argument"); return REEF_ERR_ARG; }
    if (grp->split != REEF_SPLIT_WINDOWS) { set_error("reef_msm_group_rows: every member needs the whole key (create the group with REEF_SPLIT_WINDOWS)"); return REEF_ERR_ARG; }
    if (row_len > grp->n) { set_error("row_len = %zu exceeds the key length %zu", row_len, grp->n); return REEF_ERR_ARG; }
    if (rows == 0) return REEF_OK;
    std::lock_guard<std::mutex> lk(grp->mu);
    return guarded([&]() -> reef_status {
        if (rows == 1)                                 // CE::commit (+ blind): split by window, the blind term is member 0's
            return split_call(grp, [&](size_t i, reef_jacobian *target) -> reef_status {
                Member &mb = grp->m[i];
                const void *s = scalars, *b = blinds, *hh = h;
                if (scalars_loc == REEF_DEVICE) {
                    // one staging buffer: scalars, then the blind, then h
                    if (mb.device != grp->m[0].device) {
                        const size_t sb = row_len * sizeof(reef_fe);
                        REEF_TRY(member_stage(mb, sb + sizeof(reef_fe) + sizeof(reef_affine)));
                        char *st = (char *)mb.stage;
                        if (sb) REEF_HIP_TRY(hipMemcpyPeerAsync(st, mb.device, scalars, grp->m[0].device, sb, mb.stream));
                        s = st;
                        if (blinds) {
                            REEF_HIP_TRY(hipMemcpyPeerAsync(st + sb, mb.device, blinds, grp->m[0].device, sizeof(reef_fe), mb.stream));
                            REEF_HIP_TRY(hipMemcpyPeerAsync(st + sb + sizeof(reef_fe), mb.device, h, grp->m[0].device, sizeof(reef_affine), mb.stream));
                            b = st + sb;
                            hh = st + sb + sizeof(reef_fe);
                        }
                    }
                }
                return reef_msm_rows(mb.ctx, (const reef_fe *)s, 1, row_len, scalars_loc, is_mont, max_scalar_bits, (const reef_fe *)b, (const reef_affine *)hh,
                                     target, REEF_DEVICE);
            }, out);
        return dealt_rows(grp, rows, [&](size_t i, size_t r0, size_t cnt) -> reef_status {
            Member &mb = grp->m[i];
            const void *s = scalars + r0 * row_len, *b = blinds ? blinds + r0 : nullptr, *hh = h;
            if (scalars_loc == REEF_DEVICE && mb.device != grp->m[0].device) {
                const size_t sb = cnt * row_len * sizeof(reef_fe), bb = blinds ? cnt * sizeof(reef_fe) : 0;
                REEF_TRY(member_stage(mb, sb + bb + sizeof(reef_affine)));
                char *st = (char *)mb.stage;
                hipStream_t q = (hipStream_t)reef_msm_ctx_stream(mb.whole);
                if (!q) { set_error("group member got no stream"); return REEF_ERR_HIP; }
                if (sb) REEF_HIP_TRY(hipMemcpyPeerAsync(st, mb.device, s, grp->m[0].device, sb, q));
                s = st;
                if (blinds) {
                    REEF_HIP_TRY(hipMemcpyPeerAsync(st + sb, mb.device, b, grp->m[0].device, bb, q));
                    REEF_HIP_TRY(hipMemcpyPeerAsync(st + sb + bb, mb.device, h, grp->m[0].device, sizeof(reef_affine), q));
                    b = st + sb;
                    hh = st + sb + bb;
                }
            }
            return reef_msm_rows(mb.whole, (const reef_fe *)s, cnt, row_len, scalars_loc, is_mont, max_scalar_bits, (const reef_fe *)b, (const reef_affine *)hh,
                                 out + r0, REEF_HOST);
        });
    });
}

reef_status reef_msm_group_rows_symbols(reef_msm_group *grp, const uint8_t *symbols, size_t rows, size_t row_len, int symbols_loc, uint32_t symbol_bits,
                                        const reef_fe *blinds, const reef_affine *h, bool blinds_are_mont, reef_jacobian *out) {
    if (!grp || !out || ((rows * row_len) && !symbols) || (blinds && !h)) { set_error("null argument"); return REEF_ERR_ARG; }
    if (grp->split != REEF_SPLIT_WINDOWS) { set_error("reef_msm_group_rows_symbols: every member needs the whole key (create the group with REEF_SPLIT_WINDOWS)"); return REEF_ERR_ARG; }
    if (row_len > grp->n) { set_error("row_len = %zu exceeds the key length %zu", row_len, grp->n); return REEF_ERR_ARG; }
    if (rows == 0) return REEF_OK;
    std::lock_guard<std::mutex> lk(grp->mu);
    return guarded([&]() -> reef_status {
        // one row: no windows to split on the symbol path -- member 0 computes it whole (what the engine does for rank 0 of a split)
        const size_t nd = rows == 1 ? 1 : grp->m.size();
        Outcome oc;
        auto one = [&](size_t i) {
            if (i >= nd) return;
            const size_t r0 = rows * i / nd, r1 = rows * (i + 1) / nd;
            if (r1 == r0) return;
            const size_t cnt = r1 - r0;
            Member &mb = grp->m[i];
            DeviceGuard dg(mb.device);
            if (!dg.ok) { set_error("cannot make device %d current: %s", mb.device, hipGetErrorString(dg.err)); oc.note(REEF_ERR_HIP); return; }
            oc.note([&]() -> reef_status {
                const void *s = symbols + r0 * row_len, *b = blinds ? blinds + r0 : nullptr, *hh = h;
                if (symbols_loc == REEF_DEVICE && mb.device != grp->m[0].device) {
                    const size_t sb = (cnt * row_len + 31) / 32 * 32, bb = blinds ? cnt * sizeof(reef_fe) : 0;
                    REEF_TRY(member_stage(mb, sb + bb + sizeof(reef_affine)));
                    char *st = (char *)mb.stage;
                    hipStream_t q = (hipStream_t)reef_msm_ctx_stream(mb.whole);
                    if (!q) { set_error("group member got no stream"); return REEF_ERR_HIP; }
                    if (cnt * row_len) REEF_HIP_TRY(hipMemcpyPeerAsync(st, mb.device, s, grp->m[0].device, cnt * row_len, q));
                    s = st;
                    if (blinds) {
                        REEF_HIP_TRY(hipMemcpyPeerAsync(st + sb, mb.device, b, grp->m[0].device, bb, q));
                        REEF_HIP_TRY(hipMemcpyPeerAsync(st + sb + bb, mb.device, h, grp->m[0].device, sizeof(reef_affine), q));
                        b = st + sb;
                        hh = st + sb + bb;
                    }
                }
                return reef_msm_rows_symbols(mb.whole, (const uint8_t *)s, cnt, row_len, symbols_loc, symbol_bits, (const reef_fe *)b, (const reef_affine *)hh,
                                             blinds_are_mont, out + r0, REEF_HOST);
            }());
        };
        if (grp->workers) grp->workers->run(one);
        else one(0);
        return oc.finish();
    });
}

}  // extern "C"
