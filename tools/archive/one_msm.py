#!/usr/bin/env python3
"""Run a few MSMs of one size (for rocprofv3 timelines): python tools/one_msm.py logn c G [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reef_amd import msm
logn, c, g = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
n = 1 << logn
bases = msm.gen_bases("pallas", 12345, 7, n, device=True)
sc = msm.gen_scalars("pallas", 99, n, kind=0, device=True)
out = msm.DeviceBuffer(96)
ctx = msm.MsmContext("pallas", bases, n, window_bits=c, bucket_groups=g, byte_tables=2)
ctx.enable_timing(True)
for _ in range(reps):
    ctx.msm(sc, n, out=out)
    ctx.sync()
print(ctx.plan(), ctx.timing_stats())
