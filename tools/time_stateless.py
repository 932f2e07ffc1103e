#!/usr/bin/env python3
"""Wall time of the pasta-msm drop-in symbol (host buffers: bases + scalars cross PCIe on every call).
Reports the first calls on a key (plain path; the second also builds the resident copy) and the steady
state once the key is recognised.  REEF_MSM_KEY_CACHE=0 shows the plain path throughout.

  --threads 1,4,8   callers inside the symbol at once on the SAME key (nova-snark's rayon workers, framework.rs:110,668,695):
                    per-call latency and aggregate rate, plus what the process-wide key table holds afterwards."""
import argparse
import ctypes
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reef_amd import _ffi, msm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("logn", nargs="*", type=int)
ap.add_argument("--threads", default="")
args = ap.parse_args()


def cache_info():
    st = _ffi.KeyCacheStats()
    _ffi.load().reef_key_cache_info(ctypes.byref(st))
    return st


if not args.threads:
    for logn in args.logn or [12, 14, 16, 18, 20]:
        n = 1 << logn
        bases = msm.gen_bases("pallas", 5 + logn, 3, n)
        sc = msm.gen_scalars("pallas", 9, n)
        first = []
        for _ in range(3):
            t0 = time.perf_counter()
            msm.mult_pippenger("pallas", bases, sc)
            first.append((time.perf_counter() - t0) * 1e3)
        _ffi.load().reef_key_cache_wait()              # the steady state below is the resident key's (the first calls one by one: seam_bench first=1)
        msm.mult_pippenger("pallas", bases, sc)        # the first hit takes the builder's spare context: not part of the steady state either
        tm = _ffi.KeyCacheTiming()
        _ffi.load().reef_key_cache_timing_get(None, 1)
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            msm.mult_pippenger("pallas", bases, sc)
        dt = (time.perf_counter() - t0) / reps
        _ffi.load().reef_key_cache_timing_get(ctypes.byref(tm), 0)
        if os.environ.get("REEF_MSM_LOG") == "2" and tm.calls:
            print(f"    host time per steady call (us): nominate {tm.nominate_ns/tm.calls/1e3:.1f} enqueue {tm.enqueue_ns/tm.calls/1e3:.1f} confirm {tm.confirm_ns/tm.calls/1e3:.1f} "
                  f"wait {tm.wait_ns/tm.calls/1e3:.1f} ({tm.calls} calls)", file=sys.stderr)
        print(f"mult_pippenger_pallas n=2^{logn}: calls 1-3 {first[0]:.2f} / {first[1]:.2f} / {first[2]:.2f} ms, then {dt*1e3:.3f} ms "
              f"= {n/dt/1e6:.1f} Mpairs/s (PCIe-inclusive, {96*n/dt/1e9:.2f} GB/s of host input)", flush=True)
    sys.exit(0)

counts = [int(x) for x in args.threads.split(",")]
for n in [1 << x for x in (args.logn or [])] or [3000, 27790, 1 << 16, 1 << 18]:
    bases = msm.gen_bases("pallas", 11 + n % 97, 3, n)
    scs = [msm.gen_scalars("pallas", 20 + j, n) for j in range(8)]
    for i in range(3):                                 # warm: the builder thread publishes the resident copy after the second call
        msm.mult_pippenger("pallas", bases, scs[0])
        if i == 1:
            _ffi.load().reef_key_cache_wait()
    line = [f"mult_pippenger_pallas n={n}:"]
    for nt in counts:
        reps = 20
        lat = [0.0] * nt
        barrier = threading.Barrier(nt + 1)

        def work(t):
            msm.mult_pippenger("pallas", bases, scs[t % 8])       # this thread's clone of the resident key
            barrier.wait()
            t0 = time.perf_counter()
            for _ in range(reps):
                msm.mult_pippenger("pallas", bases, scs[t % 8])
            lat[t] = (time.perf_counter() - t0) / reps
        ts = [threading.Thread(target=work, args=(t,)) for t in range(nt)]
        [t.start() for t in ts]
        barrier.wait()
        t0 = time.perf_counter()
        [t.join() for t in ts]
        wall = time.perf_counter() - t0
        line.append(f"{nt} thread{'s' if nt > 1 else ''}: {1e3*sum(lat)/nt:.3f} ms per call, {nt*reps*n/wall/1e6:.1f} Mpairs/s;")
    st = cache_info()
    line.append(f"key table: {st.resident_keys} resident keys, {st.resident_bytes/2**20:.0f} MiB, {st.builds} builds, {st.clones} clones, {st.hits} hits")
    print(" ".join(line), flush=True)
