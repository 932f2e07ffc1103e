#!/usr/bin/env python3
"""Latency of small MSMs on a resident pre-shifted key (nibble-table path): device time per call by HIP events and host
wall time per call with the result returned to the host; the 1-point commitment with a blind (CE::commit)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reef_amd import msm
h = np.ascontiguousarray(msm.gen_bases("pallas", 0xB11D, 1, 1)[0])
for n in [int(a) for a in sys.argv[1:]] or (1, 2, 16, 128, 512, 1024, 2048):
    bases = msm.gen_bases("pallas", 5, 3, n, device=True)
    sc = msm.gen_scalars("pallas", 9, n)
    dsc = msm.DeviceBuffer.from_host(sc)
    dout = msm.DeviceBuffer(96)
    with msm.MsmContext("pallas", bases, n, bucket_groups=1, byte_tables=2) as ctx:
        ctx.enable_timing(True)
        for _ in range(3): ctx.msm(dsc, n, out=dout)
        ctx.sync(); ctx.timing_stats(reset=True)
        for _ in range(20): ctx.msm(dsc, n, out=dout)
        ctx.sync()
        st = ctx.timing_stats()
        t0 = time.perf_counter()
        for _ in range(20): ctx.msm(sc)
        wall = (time.perf_counter() - t0) / 20
        line = f"n={n:5d}  device {st['total_ms'] / st['calls'] * 1e3:7.1f} us   host-to-host wall {wall * 1e6:7.1f} us"
        if n <= 2:
            b = msm.gen_scalars("pallas", 10, 1)
            ctx.msm_rows(sc, 1, n, blinds=b, h=h)
            t0 = time.perf_counter()
            for _ in range(20): ctx.msm_rows(sc, 1, n, blinds=b, h=h)
            line += f"   with blind (v*G + b*H) wall {(time.perf_counter() - t0) / 20 * 1e6:7.1f} us"
        print(line, flush=True)
