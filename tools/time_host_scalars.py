"""Wall time of reef_msm on a resident key with the scalars in host memory (what a prover that keeps
its witness on the host sees) against device-resident scalars.  Usage: python tools/time_host_scalars.py [logn ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reef_amd import msm  # noqa: E402

for logn in [int(x) for x in sys.argv[1:]] or [15, 16, 17, 20]:
    n = 1 << logn
    bases = msm.gen_bases("pallas", 5, 3, n, device=True)
    dsc = msm.gen_scalars("pallas", 9, n, device=True)
    hsc = dsc.to_host((n, 4))
    with msm.MsmContext("pallas", bases, n, bucket_groups=1) as ctx:
        res = {}
        for name, sc in (("device", dsc), ("host", hsc)):
            for _ in range(3):
                ctx.msm(sc, n)
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps):
                ctx.msm(sc, n)
            res[name] = (time.perf_counter() - t0) / reps * 1e3
        print(f"logn={logn}: scalars on device {res['device']:.3f} ms, on host {res['host']:.3f} ms "
              f"(+{res['host'] - res['device']:.3f} ms for {n * 32 / 1e6:.1f} MB = {n * 32 / 1e6 / max(res['host'] - res['device'], 1e-6):.1f} GB/s)")
