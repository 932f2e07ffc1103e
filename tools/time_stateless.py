#!/usr/bin/env python3
"""Wall time of the pasta-msm drop-in symbol (host buffers: bases + scalars cross PCIe each call)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reef_amd import msm
for logn in [int(x) for x in sys.argv[1:]] or [12, 14, 16, 18, 20]:
    n = 1 << logn
    bases = msm.gen_bases("pallas", 5, 3, n)
    sc = msm.gen_scalars("pallas", 9, n)
    msm.mult_pippenger("pallas", bases, sc)
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps):
        msm.mult_pippenger("pallas", bases, sc)
    dt = (time.perf_counter() - t0) / reps
    print(f"mult_pippenger_pallas n=2^{logn}: {dt*1e3:.3f} ms  {n/dt/1e6:.1f} Mpairs/s (PCIe-inclusive, {96*n/dt/1e9:.2f} GB/s of host input)", flush=True)
