#!/usr/bin/env python3
"""One IPA round (both cross terms) over a resident key of 2^logn generators: bucket pipeline against byte tables, and one
commitment over folded generators (reef_msm_folded).   python tools/time_ipa.py [logn ...]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from reef_amd import msm

R = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
for logn in [int(a) for a in sys.argv[1:]] or [14, 15, 16]:
    n = 1 << logn
    gens = msm.gen_bases("pallas", 5, 3, n, device=True)
    a = msm.gen_scalars("pallas", 3, n)
    for tables in (2, 1):
        with msm.MsmContext("pallas", gens, n, bucket_groups=1, byte_tables=tables) as ctx:
            for k in (0, 3):
                w1s = [(0x1234567 * (m + 3)) % R for m in range(k)]
                w2s = [(0x7654321 * (m + 5)) % R for m in range(k)]
                a_k = np.ascontiguousarray(a[: n >> k])
                ctx.ipa_cross_terms(a_k, w1s, w2s)
                best = 1e9
                for _ in range(10):
                    t = time.perf_counter(); ctx.ipa_cross_terms(a_k, w1s, w2s); best = min(best, time.perf_counter() - t)
                ctx.msm_folded(a_k, w1s, w2s)
                bf = 1e9
                for _ in range(10):
                    t = time.perf_counter(); ctx.msm_folded(a_k, w1s, w2s); bf = min(bf, time.perf_counter() - t)
                print(f"2^{logn} generators, {'byte tables' if tables == 1 else 'bucket pipeline'}, round {k}: cross terms (L and R) {best * 1e3:.3f} ms, "
                      f"one commitment over the folded generators {bf * 1e3:.3f} ms  (host to host)")
