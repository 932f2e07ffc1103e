#!/usr/bin/env python3
"""PCIe-inclusive MSM rate: scalars in pinned host memory, result to the host, T caller threads each on its own clone of the
resident key.  Also the raw upload time of the scalars.   python tools/time_host_scalars.py [logn]"""
import sys, time, threading
sys.path.insert(0, ".")
import numpy as np
import torch
from reef_amd import msm

logn = int([a for a in sys.argv[1:] if a.isdigit()][0]) if [a for a in sys.argv[1:] if a.isdigit()] else 20
n = 1 << logn
bases = msm.gen_bases("pallas", 11, 3, n, device=True)
tables = 1 if "--tables" in sys.argv else 2          # byte tables (keys of at most 2^16 points) or the bucket pipeline
ctx0 = msm.MsmContext("pallas", bases, n, bucket_groups=1, byte_tables=tables)
sc = msm.gen_scalars("pallas", 5, n)
pin = torch.empty((n, 4), dtype=torch.int64, pin_memory=True)
pin.copy_(torch.from_numpy(np.asarray(sc).view(np.int64)))
hs = pin.numpy().view(np.uint64)
dev = torch.empty((n, 4), dtype=torch.int64, device="cuda")
for _ in range(3):
    dev.copy_(pin, non_blocking=True); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    dev.copy_(pin, non_blocking=True); torch.cuda.synchronize()
up = (time.perf_counter() - t) / 10
print(f"upload of {n * 32 >> 20} MiB from pinned memory: {up * 1e3:.3f} ms = {n * 32 / up / 1e9:.1f} GB/s")
pag = np.asarray(sc).copy()
for T in (1, 2, 3, 4, 6):
    ctxs = [ctx0] + [ctx0.clone() for _ in range(T - 1)]
    outs = [np.zeros(12, dtype=np.uint64) for _ in ctxs]
    for src, name in ((hs, "pinned"), (pag, "pageable")):
        for j in range(T):
            ctxs[j].msm(src, n, out=outs[j])
        per = 8
        def work(j):
            for _ in range(per):
                ctxs[j].msm(src, n, out=outs[j])
        th = [threading.Thread(target=work, args=(j,)) for j in range(T)]
        t = time.perf_counter()
        [x.start() for x in th]; [x.join() for x in th]
        dt = (time.perf_counter() - t) / (per * T)
        print(f"threads {T} {name:8s}: {dt * 1e3:.3f} ms per MSM = {n / dt / 1e6:.0f} M pairs/s")
    for c in ctxs[1:]:
        c.close()
