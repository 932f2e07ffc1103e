#!/usr/bin/env python3
"""Debug aid for the four-wave group operations: runs reef_test_ec_op for the given ops on the inputs of test_group_law (or
on n points of another progression) and lists the lanes that differ from the oracle.
    python tools/dbg_ops.py 16 20 22 [--n 150] [--k0 3] [--plain]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from oracle import pasta_ref as cref          # noqa: E402
from oracle.pasta_oracle import CURVES       # noqa: E402
from reef_amd import _ffi, msm               # noqa: E402

args = sys.argv[1:]
n = int(args[args.index("--n") + 1]) if "--n" in args else 150
k0 = int(args[args.index("--k0") + 1]) if "--k0" in args else 3
plain = "--plain" in args
ops = [int(a) for a in args if a.isdigit() and args[max(0, args.index(a) - 1)] not in ("--n", "--k0")]
lib = _ffi.load()
C = CURVES["pallas"]
P = cref.gen_bases_ap(0, k0, 5, n)
Q = cref.gen_bases_ap(0, 1000, 9, n)
if not plain:
    Q[0] = P[0]
    Q[1] = np.frombuffer(C.affine_to_bytes(C.neg(C.affine_from_bytes(P[1].tobytes()))), dtype=np.uint64)
    Q[2] = 0; P[3] = 0; P[4] = 0; Q[4] = 0
k = np.zeros((n, 4), dtype=np.uint64)
out = np.zeros((n, 12), dtype=np.uint64)
pts = [C.affine_from_bytes(P[i].tobytes()) for i in range(n)]
qts = [C.affine_from_bytes(Q[i].tobytes()) for i in range(n)]
A, D = (lambda u, v: C.add(u, v)), (lambda u: C.add(u, u))
expect = {4: lambda p, q: A(p, q), 5: lambda p, q: D(p), 6: lambda p, q: A(D(D(A(p, q))), p), 11: lambda p, q: D(A(p, q)), 12: lambda p, q: A(A(p, q), p),
          13: lambda p, q: D(D(A(p, q))), 14: lambda p, q: A(D(p), q), 15: lambda p, q: A(D(D(p)), q), 16: lambda p, q: D(D(p)), 17: lambda p, q: D(D(p)),
          18: lambda p, q: D(D(p)), 19: lambda p, q: D(D(p)), 20: lambda p, q: D(D(p)), 21: lambda p, q: A(p, q), 22: lambda p, q: C.mul(7, A(p, q))}
for op in ops:
    assert lib.reef_test_ec_op(0, op, P.ctypes.data, Q.ctypes.data, k.ctypes.data, out.ctypes.data, n) == 0
    comp = msm.compress(0, out)
    bad = [i for i in range(n) if comp[32 * i:32 * i + 32] != C.compress(expect[op](pts[i], qts[i]))]
    print("op", op, "n", n, "bad lanes:", bad[:40], len(bad), flush=True)
