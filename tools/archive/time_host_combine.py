#!/usr/bin/env python3
"""Plain (not pre-shifted) keys with the result going to the host: the window combine sum_g 2^(c*g) S_g on a host core
(default) against the same chain of ~255 doublings on the device (REEF_MSM_HOST_COMBINE=0).  Run once per setting."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from reef_amd import msm
mode = os.environ.get("REEF_MSM_HOST_COMBINE", "1")
for logn in (12, 15, 17):
    n = 1 << logn
    bases = msm.gen_bases("pallas", 5, 3, n, device=True)
    sc = msm.gen_scalars("pallas", 9, n, device=True)
    with msm.MsmContext("pallas", bases, n, bucket_groups=0) as ctx:
        for _ in range(3): ctx.msm(sc, n)
        t0 = time.perf_counter()
        for _ in range(20): r = ctx.msm(sc, n)
        print(f"HOST_COMBINE={mode} plain key 2^{logn}: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per MSM (device scalars, result to the host), plan {ctx.plan()}", flush=True)
