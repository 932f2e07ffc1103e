#!/usr/bin/env python3
"""What a multi-GiB device allocation costs on this box: the byte tables of a 2^15 / 2^16-point key are 8.6 / 17 GB."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reef_amd import _ffi  # noqa: E402

lib = _ffi.load()
assert lib.reef_device_count() > 0
lib.reef_device_free(lib.reef_device_alloc(1 << 20))          # the first HIP call of the process
for gib in (1, 4.3, 8.6, 17.2, 8.6, 17.2):
    n = int(gib * (1 << 30))
    t0 = time.perf_counter()
    p = lib.reef_device_alloc(n)
    t1 = time.perf_counter()
    lib.reef_device_free(p)
    t2 = time.perf_counter()
    print(f"hipMalloc {gib:5.1f} GiB: {(t1 - t0) * 1e3:8.2f} ms, hipFree {(t2 - t1) * 1e3:8.2f} ms", flush=True)
