#!/usr/bin/env python3
"""Sweep (window bits, bucket groups) for several key sizes on the GPU; prints stream time per MSM.
   python tools/sweep_plans.py [logn ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reef_amd import msm

def run(logn, c, g, kind=0, reps=6, chunk=0, byte_tables=2):
    n = 1 << logn
    bases = msm.gen_bases("pallas", 12345, 7, n, device=True)
    sc = msm.gen_scalars("pallas", 99, n, kind=kind, device=True)
    out = msm.DeviceBuffer(96)
    try:
        ctx = msm.MsmContext("pallas", bases, n, window_bits=c, bucket_groups=g, chunk=chunk, byte_tables=byte_tables)
    except msm.ReefError as e:
        return None
    ctx.enable_timing(True)
    for _ in range(2):
        ctx.msm(sc, n, out=out)
    ctx.sync(); ctx.timing_stats(reset=True)
    for _ in range(reps):
        ctx.msm(sc, n, out=out)
    ctx.sync()
    st = ctx.timing_stats()
    ctx.close()
    return st["total_ms"] / st["calls"], st["accumulate_ms"] / st["calls"]

if __name__ == "__main__":
    logns = [int(x) for x in sys.argv[1:]] or [14, 16, 17, 20]
    for logn in logns:
        if 10 < logn <= 16:     # resident key with byte tables (no window to sweep: signed bytes)
            r = run(logn, 0, 1, byte_tables=1)
            print(f"## logn={logn} byte tables   {r[0]:.3f} ms  -> {(1<<logn)/r[0]/1e3:.1f} Mpairs/s   (accumulation {r[1]:.3f} ms)", flush=True)
        for g in (1, 0):
            best = None
            for c in range(max(6, logn - 8), min(20, logn + 1) + 1):
                r = run(logn, c, g)
                if r is None: continue
                print(f"logn={logn} G={'1' if g==1 else 'W'} c={c:2d} total_ms={r[0]:.3f} accum_ms={r[1]:.3f}", flush=True)
                if best is None or r[0] < best[1]: best = (c, r[0])
            print(f"## logn={logn} G={'1' if g==1 else 'W'} best c={best[0]} {best[1]:.3f} ms  -> {(1<<logn)/best[1]/1e3:.1f} Mpairs/s", flush=True)
