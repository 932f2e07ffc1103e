#!/usr/bin/env python3
"""Row N1 timing: reef_derive_generators for Reef's key sizes on the GPU, the host XOF alone, and the oracle on a sample.
    python tools/time_keygen.py [out.json]"""
import json
import sys
import time

sys.path.insert(0, ".")
from oracle import keygen_oracle as K   # noqa: E402  (timed as the CPU port, and source of the stand-in parameters)
from reef_amd import keygen             # noqa: E402

res = {"what": "CommitmentGens::new(label, n): SHAKE256 on the host + 2n hash-to-curve maps on the GPU, affine ABI points back in host memory",
       "rows": []}
k = K.standin_params("pallas")
keygen.derive_generators("pallas", b"warm", 64, k.a, k.b, k.z, k.iso, k.dst)
for logn in (10, 14, 17, 20):
    n = 1 << logn
    best = 1e9
    for _ in range(3):
        t = time.perf_counter()
        keygen.derive_generators("pallas", b"ck", n, k.a, k.b, k.z, k.iso, k.dst)
        best = min(best, time.perf_counter() - t)
    t = time.perf_counter()
    keygen.shake256(b"ck", 32 * n)
    xof = time.perf_counter() - t
    res["rows"].append({"n": n, "total_ms": round(best * 1e3, 3), "host_xof_ms": round(xof * 1e3, 3), "generators_per_s": round(n / best)})
t = time.perf_counter()
K.from_label(b"ck", 64, k)
cpu = (time.perf_counter() - t) / 64
res["oracle_python_ms_per_generator"] = round(cpu * 1e3, 3)
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
