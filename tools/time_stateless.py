#!/usr/bin/env python3
"""Wall time of the pasta-msm drop-in symbol (host buffers: bases + scalars cross PCIe on every call).
Reports the first calls on a key (plain path; the second also builds the resident copy) and the steady
state once the key is recognised.  REEF_MSM_KEY_CACHE=0 shows the plain path throughout."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reef_amd import msm  # noqa: E402

for logn in [int(x) for x in sys.argv[1:]] or [12, 14, 16, 18, 20]:
    n = 1 << logn
    bases = msm.gen_bases("pallas", 5 + logn, 3, n)
    sc = msm.gen_scalars("pallas", 9, n)
    first = []
    for _ in range(3):
        t0 = time.perf_counter()
        msm.mult_pippenger("pallas", bases, sc)
        first.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        msm.mult_pippenger("pallas", bases, sc)
    dt = (time.perf_counter() - t0) / reps
    print(f"mult_pippenger_pallas n=2^{logn}: calls 1-3 {first[0]:.2f} / {first[1]:.2f} / {first[2]:.2f} ms, then {dt*1e3:.3f} ms "
          f"= {n/dt/1e6:.1f} Mpairs/s (PCIe-inclusive, {96*n/dt/1e9:.2f} GB/s of host input)", flush=True)
