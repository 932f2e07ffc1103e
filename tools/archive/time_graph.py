#!/usr/bin/env python3
"""Host-to-host latency of one MSM on a resident pre-shifted key (bucket pipeline), the way a prover issues it: scalars in host
memory, commitment back to the host, one call at a time.  Run twice: REEF_MSM_GRAPH=0 / 1 (captured hipGraph per call shape)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reef_amd import msm  # noqa: E402

for logn in [int(x) for x in sys.argv[1:]] or [12, 14, 15, 16, 17, 18]:
    n = 1 << logn
    bases = msm.gen_bases("pallas", 12345, 7, n, device=True)
    sc = msm.gen_scalars("pallas", 99, n)
    dsc = msm.DeviceBuffer.from_host(sc)
    dout = msm.DeviceBuffer(96)
    with msm.MsmContext("pallas", bases, n, bucket_groups=1, byte_tables=2) as ctx:
        ref = msm.compress("pallas", ctx.msm(sc))
        for _ in range(4):
            assert msm.compress("pallas", ctx.msm(sc)) == ref          # the captured launches give the same point
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.msm(sc)
        host = (time.perf_counter() - t0) / reps * 1e3
        for _ in range(4):
            ctx.msm(dsc, n, out=dout)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.msm(dsc, n, out=dout)
            ctx.sync()
        dev = (time.perf_counter() - t0) / reps * 1e3
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.msm(dsc, n, out=dout)
        enq = (time.perf_counter() - t0) / reps * 1e3
        ctx.sync()
        back = (time.perf_counter() - t0) / reps * 1e3
        assert msm.compress("pallas", dout.to_host(12)) == ref
    print(f"REEF_MSM_GRAPH={os.environ.get('REEF_MSM_GRAPH', '0')} 2^{logn}: host scalars -> host result {host:.3f} ms; device scalars, sync per call {dev:.3f} ms; "
          f"back to back {back:.3f} ms per MSM (host enqueue {enq:.3f} ms)", flush=True)
