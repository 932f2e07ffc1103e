#!/usr/bin/env python3
"""Latency probes of the group operations: 256 chained additions / 64 chained doublings on ONE workgroup, four-wave
forms (ops 7, 8) against the one-wave forms (ops 9, 10); results checked, durations read from a rocprofv3 kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import pasta_ref as R
from oracle.pasta_oracle import CURVES
from reef_amd import _ffi, msm
lib = _ffi.load()
C = CURVES["pallas"]
n = 64
P = R.gen_bases_ap(0, 3, 5, n); Q = R.gen_bases_ap(0, 1000, 9, n)
k = np.zeros((n, 4), dtype=np.uint64); out = np.zeros((n, 12), dtype=np.uint64)
pts = [C.affine_from_bytes(P[i].tobytes()) for i in range(n)]; qts = [C.affine_from_bytes(Q[i].tobytes()) for i in range(n)]
for rep in range(3):
    for op, exp in ((7, lambda i: C.add(pts[i], C.mul(256, qts[i]))), (8, lambda i: C.mul(1 << 64, pts[i])),
                    (9, lambda i: C.add(pts[i], C.mul(256, qts[i]))), (10, lambda i: C.mul(1 << 64, pts[i]))):
        assert lib.reef_test_ec_op(0, op, P.ctypes.data, Q.ctypes.data, k.ctypes.data, out.ctypes.data, n) == 0
        comp = msm.compress(0, out)
        assert all(comp[32 * i:32 * i + 32] == C.compress(exp(i)) for i in (0, 17, 63)), op
print("probes ok")
