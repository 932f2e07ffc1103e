#!/usr/bin/env python3
"""Device groups on whatever devices the box has (ordinals repeat beyond them): latency of one MSM over a group against one context,
and the Hyrax rows of configs[3] dealt out over the members.  -> profiles/r05_group_timing.txt"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reef_amd import msm  # noqa: E402


def med(f, reps=9):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3, ts[0] * 1e3, ts[-1] * 1e3


def main():
    vis = msm.device_count()
    print(f"visible devices: {vis}")
    for logn in (12, 16, 20):
        n = 1 << logn
        bases = msm.gen_bases("pallas", 77, 3, n, device=True)
        dsc = msm.gen_scalars("pallas", 5, n, device=True)
        hsc = dsc.to_host((n, 4))
        with msm.MsmContext("pallas", bases, n, bucket_groups=1) as c:
            print(f"2^{logn} one context: device scalars %.3f ms (min %.3f max %.3f), host scalars %.3f ms" % (*med(lambda: c.msm(dsc, n)), med(lambda: c.msm(hsc))[0]))
        for members in (1, 2, 4, 8):
            for name, sp in (("windows", msm.SPLIT_WINDOWS), ("points", msm.SPLIT_POINTS)):
                for ex in (msm.EXCHANGE_PEER, msm.EXCHANGE_HOST, msm.EXCHANGE_RCCL):
                    with msm.MsmGroup("pallas", bases, [i % vis for i in range(members)], n, split=sp, exchange=ex) as g:
                        a = med(lambda: g.msm(dsc, n))
                        b = med(lambda: g.msm(hsc))
                        print(f"2^{logn} group of {members} by {name:7s} {g.info()['exchange']:11s}: device scalars %.3f ms (min %.3f max %.3f), host scalars %.3f ms" % (*a, b[0]))
    rows, row_len, bits = 4096, 8192, 3
    doc = np.random.default_rng(1).integers(0, 7, size=rows * row_len, dtype=np.uint8)
    hb = msm.gen_bases("pallas", 9, 3, row_len, device=True)
    with msm.MsmContext("pallas", hb, row_len) as c:
        print("hyrax rows 4096 x 8192, host bytes, one context: %.3f ms (min %.3f max %.3f)" % med(lambda: c.msm_rows_symbols(doc, rows, row_len, bits), 5))
    for members in (1, 2, 3, 4, 8):
        with msm.MsmGroup("pallas", hb, [i % vis for i in range(members)], row_len, split=msm.SPLIT_WINDOWS, bucket_groups=0) as g:
            print(f"hyrax rows over a group of {members}: %.3f ms (min %.3f max %.3f)" % med(lambda: g.msm_rows_symbols(doc, rows, row_len, bits), 5))


if __name__ == "__main__":
    main()
