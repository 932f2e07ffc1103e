#!/usr/bin/env python3
"""Single-MSM latency against the accumulation chunk length L0 (entries per k_accum0 thread).
   python tools/sweep_chunk.py logn [c ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.sweep_plans import run  # noqa: E402

if __name__ == "__main__":
    logn = int(sys.argv[1])
    cs = [int(x) for x in sys.argv[2:]] or [13, 15]
    for c in cs:
        for chunk in [int(x) for x in os.environ.get("CHUNKS", "0,8,16,32,64,128").split(",")]:
            r = run(logn, c, 1, chunk=chunk)
            if r:
                print(f"logn={logn} c={c} chunk={chunk:3d} total_ms={r[0]:.3f} accum_ms={r[1]:.3f}", flush=True)
