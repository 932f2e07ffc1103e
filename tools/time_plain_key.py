#!/usr/bin/env python3
"""The variable-base path with the result left on the device: one MSM at a time on a PLAIN key (bucket_groups = 0: no pre-shifted tables -- pasta-msm's
contract) against the same on a pre-shifted key, host wall clock around call + sync (the stream events of the context stop before the window combine's host
function).  REEF_MSM_HOST_COMBINE=0 keeps the combine on the device (the on-device Horner chain, rounds 1-5).   python tools/time_plain_key.py [logn ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reef_amd import msm  # noqa: E402

for logn in [int(x) for x in sys.argv[1:]] or [12, 14, 15, 16, 18, 20]:
    n = 1 << logn
    bases = msm.gen_bases("pallas", 12345, 7, n, device=True)
    sc = msm.gen_scalars("pallas", 99, n, device=True)
    out = msm.DeviceBuffer(96)
    row = [f"2^{logn}:"]
    ref = None
    for label, groups in (("plain key", 0), ("pre-shifted key", 1)):
        with msm.MsmContext("pallas", bases, n, bucket_groups=groups) as ctx:
            for _ in range(3):
                ctx.msm(sc, n, out=out)
            ctx.sync()
            reps = 20
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.msm(sc, n, out=out)
                ctx.sync()
            one = (time.perf_counter() - t0) / reps * 1e3
            t0 = time.perf_counter()
            for _ in range(reps):                      # enqueued back to back on ONE context (one stream): the host function of call i runs beside the kernels of call i+1?  No: stream order
                ctx.msm(sc, n, out=out)
            ctx.sync()
            chained = (time.perf_counter() - t0) / reps * 1e3
            p = ctx.plan()
            got = msm.compress("pallas", out.to_host((12,)))
            ref = ref or got
            assert got == ref, "the two keys disagree"
            row.append(f"{label} (c = {p['window_bits']}, {p['tables']} table{'s' if p['tables'] > 1 else ''}) {one:.3f} ms per MSM, {chained:.3f} back to back;")
    print(" ".join(row), flush=True)
