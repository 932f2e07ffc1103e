#!/usr/bin/env python3
"""What a multi-GiB device allocation costs on this box: the byte tables of a 2^15 / 2^16-point key are 8.6 / 17 GB."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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

# first touch: the same allocation written twice (a fill kernel through reef_memcpy's device-to-device copy of a 1 GiB pattern)
import ctypes  # noqa: E402
pat = lib.reef_device_alloc(1 << 30)
for gib in (8, 16, 8):
    n = gib << 30
    t0 = time.perf_counter()
    p = lib.reef_device_alloc(n)
    t_alloc = time.perf_counter() - t0
    times = []
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(gib):
            lib.reef_memcpy(ctypes.c_void_p(p + (k << 30)), ctypes.c_void_p(pat), 1 << 30, 1, 1)
        lib.reef_device_sync()
        times.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter()
    lib.reef_device_free(p)
    t_free = time.perf_counter() - t0
    print(f"{gib:3d} GiB: alloc {t_alloc * 1e3:.2f} ms; writing all of it: first {times[0]:.1f} ms, then {times[1]:.1f} / {times[2]:.1f} ms; free {t_free * 1e3:.2f} ms", flush=True)
