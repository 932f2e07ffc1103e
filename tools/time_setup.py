#!/usr/bin/env python3
"""What `PublicParams::setup` costs on this backend: derive a commitment key on the GPU (row N1, device-resident) and
make it a resident pre-shifted MSM key (reef_msm_ctx_create with bucket_groups = 1), warm process, per key size.
    python tools/time_setup.py [out.json]"""
import json
import sys
import time

sys.path.insert(0, ".")
from oracle import keygen_oracle as K   # noqa: E402  (stand-in parameters only)
from reef_amd import keygen, msm        # noqa: E402

k = K.standin_params("pallas")
res = {"what": "label -> device-resident generators (reef_derive_generators) -> resident pre-shifted key (reef_msm_ctx_create, G = 1)", "rows": []}
warm = keygen.derive_generators("pallas", b"w", 1 << 12, k.a, k.b, k.z, k.iso, k.dst, device=True)
msm.MsmContext("pallas", warm, 1 << 12, bucket_groups=1).close()
for logn in (10, 14, 15, 16, 17, 20):
    n = 1 << logn
    row = {"n": n}
    d_ms, c_ms = [], []
    for _ in range(3):
        t = time.perf_counter()
        g = keygen.derive_generators("pallas", b"ck", n, k.a, k.b, k.z, k.iso, k.dst, device=True)
        d_ms.append((time.perf_counter() - t) * 1e3)
        t = time.perf_counter()
        ctx = msm.MsmContext("pallas", g, n, bucket_groups=1)
        c_ms.append((time.perf_counter() - t) * 1e3)
        row["plan"] = ctx.plan()
        ctx.close()
        g.free()
    row["derive_ms"], row["key_build_ms"] = round(min(d_ms), 3), round(min(c_ms), 3)
    if 1024 < n <= 65536:                    # the same key with its byte tables built at creation (opts.byte_tables = 1)
        b_ms = []
        for _ in range(2):
            g = keygen.derive_generators("pallas", b"ck", n, k.a, k.b, k.z, k.iso, k.dst, device=True)
            t = time.perf_counter()
            ctx = msm.MsmContext("pallas", g, n, bucket_groups=1, byte_tables=1)
            b_ms.append((time.perf_counter() - t) * 1e3)
            assert ctx.has_byte_tables()
            ctx.close()
            g.free()
        row["key_build_with_byte_tables_ms"] = round(min(b_ms), 3)
    res["rows"].append(row)
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
