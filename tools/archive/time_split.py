#!/usr/bin/env python3
"""One MSM as P concurrent MSMs over P slices of the points (own stream each) + the sum of the partial results, against the MSM
in one piece: is a mid-size / large MSM that runs ALONE on the chip faster when it is split?  Device-resident scalars, a sync
per MSM (the way a prover issues them)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reef_amd import msm  # noqa: E402

for logn in [int(x) for x in sys.argv[1:]] or [15, 16, 17, 18, 19, 20]:
    n = 1 << logn
    k0, d = 12345, 7
    sc = msm.gen_scalars("pallas", 99, n, device=True)
    full = msm.MsmContext("pallas", msm.gen_bases("pallas", k0, d, n, device=True), n, bucket_groups=1, byte_tables=2)
    out = msm.DeviceBuffer(96)
    reps = 40
    for _ in range(4):
        full.msm(sc, n, out=out); full.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        full.msm(sc, n, out=out); full.sync()
    t_full = (time.perf_counter() - t0) / reps * 1e3
    ref = msm.compress("pallas", out.to_host(12))
    line = f"2^{logn}: one piece {t_full:.3f} ms"
    for parts in (2, 3, 4):
        m = n // parts
        bounds = [(p * m, n if p == parts - 1 else (p + 1) * m) for p in range(parts)]
        ctxs = [msm.MsmContext("pallas", msm.gen_bases("pallas", k0 + lo * d, d, hi - lo, device=True), hi - lo, bucket_groups=1, byte_tables=2) for lo, hi in bounds]
        partial = msm.DeviceBuffer(96 * parts)
        def run():
            for p, (lo, hi) in enumerate(bounds):
                ctxs[p].msm(sc.ptr + 32 * lo, hi - lo, out=partial.ptr + 96 * p)
            for c in ctxs[1:]:
                c.sync()
            ctxs[0].sum_points(partial, parts, out)
            ctxs[0].sync()
        for _ in range(4):
            run()
        assert msm.compress("pallas", out.to_host(12)) == ref
        t0 = time.perf_counter()
        for _ in range(reps):
            run()
        t = (time.perf_counter() - t0) / reps * 1e3
        line += f"; {parts} slices {t:.3f} ms"
        for c in ctxs:
            c.close()
    print(line, flush=True)
    full.close()
