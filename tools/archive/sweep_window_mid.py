#!/usr/bin/env python3
"""Where the window size should change between 2^15 and 2^19 points (pre-shifted key, one bucket set): latency of ONE MSM in flight (stream time,
median of 9) and throughput with four in flight, for c = 12..18 at sizes between the powers of two.  python tools/sweep_window_mid.py [n ...]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from reef_amd import msm


def one(n, c, reps=9):
    bases = msm.gen_bases("pallas", 12345, 7, n, device=True)
    sc = msm.gen_scalars("pallas", 99, n, kind=0, device=True)
    out = msm.DeviceBuffer(96)
    ctx = msm.MsmContext("pallas", bases, n, window_bits=c, bucket_groups=1, byte_tables=2)
    for _ in range(3):
        ctx.msm(sc, n, out=out)
    ctx.sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ctx.msm(sc, n, out=out)
        ctx.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    cs = [ctx] + [ctx.clone() for _ in range(3)]
    outs = [msm.DeviceBuffer(96) for _ in cs]
    def work(j):
        for _ in range(24):
            cs[j].msm(sc, n, out=outs[j])
        cs[j].sync()
    for j in range(4):
        work(j)
    th = [threading.Thread(target=work, args=(j,)) for j in range(4)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    four = (time.perf_counter() - t0) / 96 * 1e3
    for c_ in cs:
        c_.close()
    return ts[len(ts) // 2], ts[0], four


if __name__ == "__main__":
    sizes = [int(x) for x in sys.argv[1:]] or [32768, 40000, 49152, 57344, 65536, 81920, 98304, 131072, 163840, 196608, 262144, 393216, 524288]
    for n in sizes:
        row = []
        for c in range(int(os.environ.get("CMIN", "12")), int(os.environ.get("CMAX", "18")) + 1):
            try:
                med, best, four = one(n, c)
            except msm.ReefError:
                continue
            row.append((c, med, four))
            print(f"n={n:7d} c={c:2d} host-to-host median {med:.3f} ms (min {best:.3f}); four in flight {four:.3f} ms per MSM", flush=True)
        bl = min(row, key=lambda r: r[1]); bt = min(row, key=lambda r: r[2])
        shipped = msm.plan_for(n, bucket_groups=1)["window_bits"]
        print(f"## n={n}: latency best c={bl[0]} ({bl[1]:.3f} ms), throughput best c={bt[0]} ({bt[2]:.3f} ms); shipped c={shipped}", flush=True)
