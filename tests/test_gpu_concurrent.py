"""The one seam an unmodified Reef reaches -- pasta-msm's `mult_pippenger_pallas/vesta` -- under the callers it really has:
nova-snark calls it from the main prover thread and from rayon workers (SURVEY.md 8b "Threading";
src/backend/framework.rs:110, 668, 695), so several threads are inside the symbol at once, on the same commitment key and on
different ones, and worker threads come and go.  Every result is compared with the C oracle (oracle/pasta_ref.c)."""
import ctypes
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SIZES = (128, 3000, 27790, 1 << 16)          # below the cache threshold, small, cfg3's W key length (replay_shapes.json), Reef's 2^16


def cache_info():
    """The table's counters, once the builder thread has nothing left to do: a build an EARLIER test handed over must not land inside the window of the test that
    reads a delta (round 6: builds are asynchronous)."""
    from reef_amd import _ffi
    _ffi.load().reef_key_cache_wait()
    st = _ffi.KeyCacheStats()
    _ffi.load().reef_key_cache_info(ctypes.byref(st))
    return {n: getattr(st, n) for n, _ in st._fields_}


def run_threads(nthreads, fn):
    errs = []

    def wrap(t):
        try:
            fn(t)
        except BaseException as e:          # an assertion in a worker must fail the test
            errs.append((t, repr(e)))
    ts = [threading.Thread(target=wrap, args=(t,)) for t in range(nthreads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs[:3]


@pytest.fixture(scope="module")
def cases(cref):
    """Per curve and size: one key, four scalar vectors (uniform and witness-like) and the oracle's compressed commitments."""
    out = {}
    for cid in (0, 1):
        for n in SIZES:
            bases = cref.gen_bases_ap(cid, 7000 + 31 * n % 997 + cid, 3, n)
            bases[n // 3] = 0                                                  # an identity base inside every key
            scal = [cref.gen_scalars(cid, 900 + 7 * j + cid, n, kind=j % 2) for j in range(4)]
            want = [cref.compress(cid, cref.msm_pippenger(cid, bases, s, threads=8)) for s in scal]
            out[(cid, n)] = (bases, scal, want)
    return out


def test_eight_threads_share_one_key(gpu_lib, cref, cases):
    """Eight threads at once on the SAME key, for every size and both curves; the threads are started and joined three times so
    that thread-local state (streams, workspaces, clones of the resident key) is torn down and rebuilt while the process-wide
    table lives on.  One resident copy per key is built, whatever the number of callers."""
    from reef_amd import msm
    gpu_lib.reef_key_cache_clear()
    before = cache_info()
    for cid in (0, 1):
        for n in SIZES:
            bases, scal, want = cases[(cid, n)]
            for generation in range(3):
                def work(t):
                    for rep in range(4):
                        j = (t + rep + generation) % 4
                        got = msm.compress(cid, msm.mult_pippenger(cid, bases, scal[j]))
                        assert got == want[j], (cid, n, generation, t, rep)
                run_threads(8, work)
                gpu_lib.reef_key_cache_wait()           # round 6: the resident copy is built beside the callers; generations 1 and 2 find it published
    after = cache_info()
    cached_keys = 2 * sum(1 for n in SIZES if n >= 1024)
    assert after["builds"] - before["builds"] == cached_keys, (before, after)       # one build per key, not one per thread
    assert after["resident_keys"] == cached_keys
    assert after["hits"] - before["hits"] >= cached_keys * 2 * 8 * 4, (before, after)   # at least every call of generations 1 and 2
    per_key = {n: 64 * n * msm.plan_for(n, bucket_groups=1)["tables"] for n in SIZES if n >= 1024}    # the bytes for the comparison stay on the host up to 2^17 points
    # one copy of each key, 16 callers or not (a delta: keys a thread of an earlier test is still attached to stay charged until it lets go)
    assert after["resident_bytes"] - before["resident_bytes"] == 2 * sum(per_key.values())
    assert after["misspeculated"] == before["misspeculated"]


def test_threads_on_different_keys_and_curves_at_once(gpu_lib, cref, cases):
    """Eight threads inside the symbol at once, each on its own (curve, size) -- both curves, all four sizes -- then every thread
    walks through all of them; more distinct keys in flight than one thread keeps clones of."""
    from reef_amd import msm
    combos = [(cid, n) for cid in (0, 1) for n in SIZES]
    for generation in range(2):
        def work(t):
            for step in range(len(combos)):
                cid, n = combos[(t + step) % len(combos)]
                bases, scal, want = cases[(cid, n)]
                j = (t + step) % 4
                assert msm.compress(cid, msm.mult_pippenger(cid, bases, scal[j])) == want[j], (generation, t, cid, n)
        run_threads(8, work)


@pytest.mark.parametrize("n", [5000, (1 << 17) + 1000])
def test_keys_that_differ_in_one_unsampled_point(n, gpu_lib, cref):
    """The fast nomination looks at 64 sampled points; two keys that agree on all of them (one point edited in place) are told
    apart by the comparison of every byte that runs beside the speculative MSM -- always on the host (round 5), by the caller alone up
    to 4 MiB of key and together with the library's helper threads above (the second size) -- and each call returns the commitment of the key it was given, from several threads alternating between them."""
    from reef_amd import msm
    cid = 1
    a = cref.gen_bases_ap(cid, 31337, 3, n)
    b = a.copy()
    b[n // 2 + 1] = cref.gen_bases_ap(cid, 99, 1, 1)[0]       # not one of the sampled indices (multiples of n // 63, and n - 1)
    assert (n // 2 + 1) % max(1, n // 63) != 0
    sc = cref.gen_scalars(cid, 5, n)
    want = {0: cref.compress(cid, cref.msm_pippenger(cid, a, sc, threads=8)), 1: cref.compress(cid, cref.msm_pippenger(cid, b, sc, threads=8))}
    assert want[0] != want[1]
    before = cache_info()
    for which in (0, 1, 0, 1):                                 # both keys seen twice: the builder thread makes them resident
        assert msm.compress(cid, msm.mult_pippenger(cid, (a, b)[which], sc)) == want[which]
    gpu_lib.reef_key_cache_wait()

    def work(t):
        for rep in range(8):
            which = (t + rep) % 2
            assert msm.compress(cid, msm.mult_pippenger(cid, (a, b)[which], sc)) == want[which], (t, rep)
    run_threads(4, work)
    gpu_lib.reef_key_cache_wait()
    after = cache_info()
    # round 6: the samples are the only hash, so the two keys are told apart by their bytes alone -- the first call that finds the OTHER key resident
    # speculates on it and is caught by the comparison; once both are resident the bytes choose BEFORE an MSM is spent
    assert after["builds"] - before["builds"] == 2 and after["misspeculated"] > before["misspeculated"]
    mid = cache_info()
    run_threads(4, work)
    assert cache_info()["misspeculated"] == mid["misspeculated"] and cache_info()["builds"] == mid["builds"]


def test_attach_moves_a_handle_between_keys(gpu_lib, cref):
    """reef_msm_ctx_attach: one stream and workspace serving several resident keys in turn (what the drop-in symbols do per thread)."""
    from reef_amd import msm
    cid = 0
    keys = [cref.gen_bases_ap(cid, 600 + 7 * j, 3, n) for j, n in enumerate((700, 5000, 40000))]      # nibble tables, c = 13 plans
    scs = [cref.gen_scalars(cid, 60 + j, len(kb)) for j, kb in enumerate(keys)]
    want = [cref.compress(cid, cref.msm_pippenger(cid, kb, sc, threads=4)) for kb, sc in zip(keys, scs)]
    owners = [msm.MsmContext(cid, kb, bucket_groups=1) for kb in keys]
    h = owners[0].clone()
    for rnd in range(3):
        for j in (2, 0, 1, 1, 2):
            h.attach(owners[j])
            assert msm.compress(cid, h.msm(scs[j])) == want[j], (rnd, j)
    owners[2].close()                                 # the handle keeps the key it is attached to alive
    assert msm.compress(cid, h.msm(scs[2])) == want[2]
    with pytest.raises(msm.ReefError):
        other = msm.MsmContext(1, cref.gen_bases_ap(1, 1, 1, 8))
        h.attach(other)                               # another curve
    h.close()
    for o in owners[:2]:
        o.close()


def test_msm_multi_matches_the_calls_one_by_one(gpu_lib, cref):
    """reef_msm_multi: the four commitments of a folding step (two per curve, the second of a curve on a clone of its key) enqueued
    together; same points as one by one, for lengths shorter than the keys; duplicate contexts are refused."""
    from reef_amd import msm
    keys = {cid: cref.gen_bases_ap(cid, 800 + cid, 3, 1 << 15) for cid in (0, 1)}
    ctx = {cid: msm.MsmContext(cid, keys[cid], bucket_groups=1) for cid in (0, 1)}
    clones = {cid: ctx[cid].clone() for cid in (0, 1)}
    lens = [(1, 11376), (0, 27790), (0, 27789), (1, 3)]
    scs = [cref.gen_scalars(cid, 70 + j, n, kind=j % 2) for j, (cid, n) in enumerate(lens)]
    want = [cref.compress(cid, cref.msm_pippenger(cid, keys[cid][:n].copy(), sc, threads=4)) for (cid, n), sc in zip(lens, scs)]
    order = [ctx[1], ctx[0], clones[0], clones[1]]
    for rep in range(3):
        got = msm.msm_multi(order, scs)
        assert [msm.compress(cid, got[j]) for j, (cid, _) in enumerate(lens)] == want, rep
    assert msm.compress(0, msm.msm_multi([clones[0]], [scs[1]])[0]) == want[1]
    with pytest.raises(msm.ReefError):
        msm.msm_multi([ctx[0], ctx[0]], [scs[1], scs[2]])
    for c_ in list(clones.values()) + list(ctx.values()):
        c_.close()


def test_contexts_whose_stream_is_handed_out_get_streams_of_their_own(gpu_lib, cref):
    """reef_msm_ctx_stream pins a context to a stream of the library's pool (RCCL / torch order their work after it): the three
    contexts a bench rank keeps in flight must not end up on ONE stream, and a pinned context keeps its stream across calls."""
    from reef_amd import msm
    bases = cref.gen_bases_ap(0, 5, 3, 4096)
    sc = cref.gen_scalars(0, 6, 4096)
    want = cref.compress(0, cref.msm_pippenger(0, bases, sc, threads=4))
    with msm.MsmContext(0, bases, bucket_groups=1) as ctx:
        clones = [ctx.clone() for _ in range(2)]
        streams = [c.stream for c in [ctx] + clones]
        assert len(set(streams)) == 3 and all(streams)
        for c, s in zip([ctx] + clones, streams):
            assert msm.compress(0, c.msm(sc)) == want
            assert c.stream == s
        for c in clones:
            c.close()


def test_table_turnover_under_concurrency(gpu_lib, cref):
    """More keys than the process-wide table has entries (16), revisited by four threads at once: entries are evicted while other
    threads still hold clones of them; results stay right and the evicted keys' memory is given back."""
    from reef_amd import msm
    cid, n = 0, 1500
    keys = [cref.gen_bases_ap(cid, 4000 + 17 * k, 5, n) for k in range(20)]
    sc = cref.gen_scalars(cid, 77, n)
    want = [cref.compress(cid, cref.msm_pippenger(cid, kb, sc, threads=4)) for kb in keys]
    gpu_lib.reef_key_cache_clear()
    base = cache_info()["resident_bytes"]           # what this (the main) thread's own attachments pin: it makes no call until the end

    def work(t):
        for rnd in range(3):
            for k in range(len(keys)):
                kk = (k + 5 * t) % len(keys)
                assert msm.compress(cid, msm.mult_pippenger(cid, keys[kk], sc)) == want[kk], (t, rnd, kk)
    run_threads(4, work)
    info = cache_info()
    assert info["entries"] <= 16 and info["resident_keys"] <= 16
    gpu_lib.reef_key_cache_clear()
    # the workers have ended: their contexts went to the builder's pool WITH the keys they were attached to, and pooled contexts of keys that left
    # the table are destroyed (round 6) -- nothing but the main thread's own attachments may stay charged
    import time
    for _ in range(300):                            # Thread.join() returns before the native thread has run its thread-local destructors
        if cache_info()["resident_bytes"] == base:
            break
        time.sleep(0.01)                            # (a context that reaches the builder after the clear is not pooled: its key has left the table)
    assert cache_info()["resident_bytes"] == base
    msm.mult_pippenger(cid, keys[0], sc)            # this thread lets go of the clones it holds of evicted keys
    assert cache_info()["entries"] == 1


def test_an_evicted_key_stays_charged_while_a_thread_is_attached_to_it(gpu_lib, cref):
    """the budget is released when the LAST handle on a key's tables goes, not when the entry leaves the table -- a thread's
    context attached to an evicted key still pins its device memory, and REEF_MSM_KEY_CACHE_MB must account for it."""
    from reef_amd import msm
    cid, n = 0, 2000
    a, b = cref.gen_bases_ap(cid, 9100, 3, n), cref.gen_bases_ap(cid, 9200, 3, n)
    sc = cref.gen_scalars(cid, 3, n)
    per = 64 * n * msm.plan_for(n, bucket_groups=1)["tables"]
    def three_calls(key):
        for i in range(3):
            msm.mult_pippenger(cid, key, sc)           # the second call hands the key to the builder thread; the third finds it resident
            if i == 1:
                gpu_lib.reef_key_cache_wait()
    three_calls(b)                                     # this thread's context is attached to b
    gpu_lib.reef_key_cache_clear()
    r0 = cache_info()
    assert r0["entries"] == 0 and r0["resident_keys"] == 0 and r0["resident_bytes"] >= per       # b left the table but is still pinned, and still charged
    three_calls(a)                                     # the thread lets go of b at its next call (the table's epoch): b's tables are freed, a's are charged
    r1 = cache_info()
    assert r1["resident_bytes"] == r0["resident_bytes"] and r1["resident_keys"] == 1
    gpu_lib.reef_key_cache_clear()
    assert cache_info()["resident_bytes"] == r0["resident_bytes"] and cache_info()["entries"] == 0
    t = threading.Thread(target=lambda: three_calls(b))     # another thread brings b back and ends: its attachment goes with it
    t.start()
    t.join()
    r2 = cache_info()
    assert r2["resident_bytes"] == r0["resident_bytes"] + per and r2["resident_keys"] == 1      # a (pinned by this thread) + b (in the table)
    gpu_lib.reef_key_cache_clear()
    import time
    for _ in range(200):                               # Thread.join() returns before the native thread has run its thread-local destructors
        if cache_info()["resident_bytes"] == r0["resident_bytes"]:
            break
        time.sleep(0.01)
    assert cache_info()["resident_bytes"] == r0["resident_bytes"]                                # b had no other holder: freed with its entry


@pytest.mark.parametrize("n", [4096, 27790])
def test_first_calls_on_a_fresh_key(n, gpu_lib, cref):
    """what a proof with 1-6 folding steps sees.  Six calls on a key the process has never seen, every result against the
    C oracle whichever path served it (plain while the builder thread works, resident once it has published the key); other threads call on
    the same key WHILE it is being built; exactly one resident copy is built and the builder's spare context serves the first hit."""
    from reef_amd import msm
    cid = 0
    bases = cref.gen_bases_ap(cid, 424242 + n, 11, n)
    scal = [cref.gen_scalars(cid, 300 + j, n, kind=j % 2) for j in range(3)]
    want = [cref.compress(cid, cref.msm_pippenger(cid, bases, s, threads=8)) for s in scal]
    before = cache_info()
    for i in range(2):                                                         # the second appearance hands the key to the builder thread
        assert msm.compress(cid, msm.mult_pippenger(cid, bases, scal[i % 3])) == want[i % 3], i

    def work(t):                                                               # beside the build: served on the plain path, or on the fresh copy
        for rep in range(3):
            j = (t + rep) % 3
            assert msm.compress(cid, msm.mult_pippenger(cid, bases, scal[j])) == want[j], (t, rep)
    run_threads(4, work)
    for i in range(2, 6):
        assert msm.compress(cid, msm.mult_pippenger(cid, bases, scal[i % 3])) == want[i % 3], i
    gpu_lib.reef_key_cache_wait()
    assert msm.compress(cid, msm.mult_pippenger(cid, bases, scal[0])) == want[0]
    after = cache_info()
    assert after["builds"] - before["builds"] == 1, (before, after)
    assert after["spares"] - before["spares"] == 1 and after["hits"] > before["hits"], (before, after)
    assert after["misspeculated"] == before["misspeculated"]


def test_runtime_warm_up_in_the_background(gpu_lib):
    """reef_runtime_init({warm = REEF_WARM_BACKGROUND}) / REEF_MSM_WARM=1: a thread of the library pays the HIP runtime's one-off costs (initialisation, the first
    stream, both curves' code objects, the copy engines, every pipeline kernel's first launch) while the prover does its host-side start-up.  Fresh processes
    (reef_amd/_lib/seam_bench cold=...): same points with and without, the warm-up reports done, and with 300 ms of host work to hide behind the first
    commitment is at least 5x sooner (measured: 1.4 ms against 94 ms, profiles/r06_seam_cold_process.txt)."""
    import json
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "reef_amd", "_lib", "seam_bench")
    with tempfile.TemporaryDirectory() as td:
        data = os.path.join(td, "cold.bin")
        subprocess.run([exe, f"gen={data}", "n=27790"], check=True, timeout=120)

        def run(warm, env=None):
            out = subprocess.run([exe, f"cold={data}", "n=27790", f"warm={warm}", "host_ms=300"], capture_output=True, text=True, timeout=120,
                                 env=dict(os.environ, **(env or {})))
            assert out.returncode == 0, out.stderr[-1500:]
            return json.loads(out.stdout.strip().splitlines()[-1])
        cold, warm, at_load = run(0), run(2), run(0, {"REEF_MSM_WARM": "1"})
    for line in (cold, warm, at_load):
        assert line["results_identical"] is True
    assert cold["warm_state_afterwards"] == 0 and warm["warm_state_afterwards"] == 2 and at_load["warm_state_afterwards"] == 2
    assert warm["call_ms"][0] * 5 < cold["call_ms"][0] and at_load["call_ms"][0] * 5 < cold["call_ms"][0], (cold, warm, at_load)
