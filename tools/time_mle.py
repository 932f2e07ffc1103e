"""Times row N3 (reef_mle_bound_rows) at BASELINE.json document sizes, device-resident tables.
Usage: python tools/time_mle.py  -> JSON lines (run under rocprofv3 for kernel durations)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reef_amd import _ffi, mle, msm  # noqa: E402
from reef_amd.sumcheck import ints_to_array  # noqa: E402

Q = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
rng = np.random.default_rng(1)
cases = [("cfg3 1MiB ascii u8", 21, 10, 1, 131), ("cfg4 16MiB dna u8", 25, 12, 1, 7), ("cfg4 hybrid table 32B", 25, 12, 32, 0),
         ("cfg5 64MiB utf8 u16", 27, 13, 2, 259), ("2^21 table 32B", 21, 10, 32, 0)]
for name, m, left, eb, bound in cases:
    n = 1 << m
    if eb == 32:
        z = msm.gen_scalars("pallas", 7, n, device=True)      # uniform field elements (Montgomery form)
        dz, is_mont = z, True
    else:
        host = rng.integers(0, bound, size=n, dtype={1: np.uint8, 2: np.uint16}[eb])
        dz, is_mont = msm.DeviceBuffer.from_host(host.view(np.uint8)), False
    point = ints_to_array([int.from_bytes(rng.bytes(31), "little") % Q for _ in range(m)])
    best = 1e9
    for it in range(6):
        t0 = time.perf_counter()
        mle.bound_rows_raw("pallas", dz, point, left, is_mont=is_mont, n=n, elem_bytes=eb)
        dt = time.perf_counter() - t0
        if it:
            best = min(best, dt)
    print(json.dumps({"case": name, "entries": n, "elem_bytes": eb, "wall_ms": round(best * 1e3, 3),
                      "table_GBps": round(n * eb / best / 1e9, 1), "entries_per_s": round(n / best / 1e9, 2)}))
    del dz
