#!/usr/bin/env python3
"""Where does one folding step of the sum-check go?  A table shaped like cfg4's hybrid table (2^ell entries), host-timed per call:
gen_eq_table, round-1 sums, then every fused round.  python tools/step_breakdown.py [ell]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reef_amd import msm
from reef_amd.sumcheck import SumCheck
Q = msm.PALLAS_SCALAR_Q
ell = int(sys.argv[1]) if len(sys.argv) > 1 else 26
n = 1 << ell
cols = 1 << (ell // 2)
hyb = np.zeros((n, 4), dtype=np.uint64)
rngn = np.random.default_rng(7)
hyb[:3 * cols + 17] = rngn.integers(0, 1 << 62, size=(3 * cols + 17, 4), dtype=np.uint64)
hyb[3 * cols + 17:n // 2] = np.array([0x123456789abcdef1, 0x0fedcba987654321, 0x1111111122222222, 0x0333333344444444], dtype=np.uint64)
hyb[n // 2:n - n // 16, 0] = rngn.integers(0, 7, size=n // 2 - n // 16, dtype=np.uint64)
d_hyb = msm.DeviceBuffer.from_host(hyb)
del hyb
rs = [(0x1234567 * (k + 3)) % Q for k in range(34)]
qs = [(0x9E3779B1 * (k + 1)) % n for k in range(33)]
lq = [(0x7654321 * (k + 5)) % Q for k in range(ell)]
with SumCheck("pallas", ell) as sc:
    sc.set_table_device(0, d_hyb.ptr, n)
    best = None
    for rep in range(4):
        sc.reset_table(); sc.sync()
        t = []
        t0 = time.perf_counter(); sc.gen_eq_table(rs, qs, lq); t.append(time.perf_counter() - t0)
        t0 = time.perf_counter(); xsq, x, con = sc.round_coeffs(1); t.append(time.perf_counter() - t0)
        for i in range(1, ell + 1):
            rch = (xsq * 7 + 3) % Q
            t0 = time.perf_counter()
            if i < ell:
                xsq, x, con = sc.fold_and_next_coeffs(i, rch)
            else:
                sc.fold(i, rch); sc.sync()
            t.append(time.perf_counter() - t0)
        if best is None or sum(t) < sum(best):
            best = t
    print(f"ell={ell}: step {sum(best) * 1e3:.3f} ms = gen_eq_table {best[0] * 1e6:.0f} us + round-1 sums {best[1] * 1e6:.0f} us + rounds (pairs 2^{ell - 1} .. 1): "
          + " ".join(f"{v * 1e6:.0f}" for v in best[2:]) + " us")
    big = sum(v for v in best[2:] if v > 60e-6)
    print(f"   rounds above 60 us: {big * 1e3:.3f} ms; the others: {(sum(best[2:]) - big) * 1e3:.3f} ms over {sum(1 for v in best[2:] if v <= 60e-6)} rounds")
