#!/usr/bin/env python3
"""Generator fold G'_i = w1*G_i + w2*G_{i+half} (reef_fold, row K3) timed device to device for a range of sizes.
usage: python tools/time_fold.py [logn ...]   (REEF_MSM_FOLD_COOP=0: the one-wave kernel)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from reef_amd import msm
from reef_amd.msm import PALLAS


def main():
    logs = [int(a) for a in sys.argv[1:]] or [8, 12, 14, 15, 16, 17]
    w1 = (0x1234567 * msm.SCALAR_MODULUS[msm.PALLAS] // 0x7654321) % msm.SCALAR_MODULUS[msm.PALLAS]
    w2 = pow(w1, -1, msm.SCALAR_MODULUS[msm.PALLAS])
    for logn in logs:
        n = 1 << logn
        gens = msm.gen_bases(msm.PALLAS, 3, 7, n, device=True)
        out = msm.fold(msm.PALLAS, gens, n // 2, w1, w2)
        msm.device_sync()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            out = msm.fold(msm.PALLAS, gens, n // 2, w1, w2, out=out)
        msm.device_sync()
        print("fold of 2^%d generators -> 2^%d: %.3f ms" % (logn, logn - 1, (time.perf_counter() - t0) / reps * 1e3))


if __name__ == "__main__":
    main()
