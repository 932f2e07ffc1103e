#!/usr/bin/env python3
"""How long does ONE Poseidon permutation take on the GPU (a latency chain on one workgroup), against the host round trip it would
replace if the sum-check's Fiat-Shamir challenge were squeezed on the device?  A document of 2 symbols is one
leaf hash: one permutation (five lanes per hash below 16384 nodes, a thread per node with REEF_POSEIDON_SPREAD=0); 4 symbols: two
dependent permutations.  Beside it: what a small sum-check round costs host to host today (launch, kernel, polled result)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes
import numpy as np
from oracle import merkle_oracle as M
from reef_amd import _ffi, merkle
from reef_amd.sumcheck import SumCheck, ints_to_array
lib = _ffi.load()
p = M.standin_params()
rc = ints_to_array(p.rc); mds = ints_to_array([x for r in p.mds for x in r])
pp = merkle.PoseidonParams(5, p.rf, p.rp, 0, rc.ctypes.data, mds.ctypes.data, (ctypes.c_uint64 * 4)(p.tag_leaf & (2**64 - 1), p.tag_leaf >> 64, 0, 0),
                           (ctypes.c_uint64 * 4)(p.tag_node & (2**64 - 1), p.tag_node >> 64, 0, 0))
res = {}
for n in (2, 4, 8):
    doc = np.arange(n, dtype=np.uint32)
    root = np.zeros((1, 4), dtype=np.uint64)
    best = 1e9
    for rep in range(30):
        t0 = time.perf_counter()
        assert lib.reef_merkle_commit(0, ctypes.byref(pp), doc.ctypes.data, n, 0, False, None, 0, root.ctypes.data) == 0
        best = min(best, time.perf_counter() - t0)
    res[n] = best
    print(f"{n} symbols ({merkle.nodes(n)} hashes, {n.bit_length() - 1} dependent permutations): {best * 1e6:.1f} us host to host (upload, launches, root back)")
print(f"=> one more dependent permutation costs {(res[4] - res[2]) * 1e6:.1f} us (4 against 2 symbols), {(res[8] - res[4]) * 1e6:.1f} us (8 against 4)")
Q = M.Q
ell = 14
with SumCheck("pallas", ell) as sc:
    sc.set_table(0, [(i * 7 + 1) % 131 for i in range(1 << ell)])
    sc.gen_eq_table([5, 6, 7], [1, 2], [(11 * k + 3) % Q for k in range(ell)])
    sc.round_coeffs(1)
    t = []
    for i in range(1, ell):
        t0 = time.perf_counter()
        sc.fold_and_next_coeffs(i, (0x1234567 * i) % Q)
        t.append(time.perf_counter() - t0)
    print(f"small sum-check rounds (2^{ell} entries and below), fold + next coefficients, host to host: median {sorted(t)[len(t) // 2] * 1e6:.1f} us, min {min(t) * 1e6:.1f} us")
    print("   per round, pow = 2^13 .. 2: " + " ".join(f"{v * 1e6:.0f}" for v in t) + f" us; sum {sum(t) * 1e6:.0f} us")
