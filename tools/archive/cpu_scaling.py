#!/usr/bin/env python3
"""How many host threads actually help the CPU baseline on this box: the oracle's window-parallel Pippenger at 1, 2, 4 ... threads,
next to what the container is allowed (nproc, cgroup quota, thread limits)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import pasta_ref as R  # noqa: E402

print("os.cpu_count:", os.cpu_count(), " sched_getaffinity:", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/pids.max"):
    try:
        print(f, "=", open(f).read().strip())
    except OSError:
        pass
try:
    model = [l for l in open("/proc/cpuinfo") if l.startswith("model name")]
    print(model[0].strip(), "x", len(model))
except OSError:
    pass
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
b = R.gen_bases_ap(0, 3, 5, n)
s = R.gen_scalars(0, 0x5EEF, n)
t = 1
base = None
while t <= (os.cpu_count() or 1):
    R.msm_pippenger_windows(0, b, s, threads=t)
    t0 = time.perf_counter()
    R.msm_pippenger_windows(0, b, s, threads=t)
    dt = time.perf_counter() - t0
    base = base or dt
    print(f"threads {t:4d} (pool holds {R.pool_size()} helpers): {n / dt / 1e6:8.3f} M pairs/s, speed-up {base / dt:6.1f}, plan c,slices = {R.window_plan(n, t)}", flush=True)
    t *= 2
