#!/usr/bin/env python3
"""One MSM configuration, repeated, for rocprofv3 --kernel-trace: per-kernel timeline of a single MSM.
   python tools/profile_one.py run LOGN C G [REPS]          # the workload (run it under rocprofv3 --kernel-trace)
   python tools/profile_one.py show TRACE.csv [SKIP]        # timeline of the last MSM in the trace (durations and gaps, us)
"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(logn, c, g, reps=8, kind=0, chunk=0):
    from reef_amd import msm
    n = 1 << logn
    bases = msm.gen_bases("pallas", 12345, 7, n, device=True)
    sc = msm.gen_scalars("pallas", 99, n, kind=kind, device=True)
    out = msm.DeviceBuffer(96)
    ctx = msm.MsmContext("pallas", bases, n, window_bits=c, bucket_groups=g, chunk=chunk, byte_tables=int(os.environ.get("BYTE_TABLES", "2")))
    ctx.enable_timing(True)
    for _ in range(3):
        ctx.msm(sc, n, out=out)
    ctx.sync()
    ctx.timing_stats(reset=True)
    for _ in range(reps):
        ctx.msm(sc, n, out=out)
        ctx.sync()
    st = ctx.timing_stats()
    print(f"logn={logn} c={c} G={g} chunk={chunk} plan={ctx.plan()} total_ms={st['total_ms'] / st['calls']:.4f} accum_ms={st['accumulate_ms'] / st['calls']:.4f}")


def show(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last MSM = from the last k_recode on
    last = max(i for i, r in enumerate(rows) if "recode" in r["Kernel_Name"])
    seq = rows[last:]
    t0 = int(seq[0]["Start_Timestamp"])
    prev_end = t0
    tot_busy = 0
    for r in seq:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void reef::", "").replace("reef::", "")
        print(f"  {name:34s} start={(s - t0) / 1e3:8.1f} dur={(e - s) / 1e3:7.1f} gap={(s - prev_end) / 1e3:6.1f}  grid={r.get('Grid_Size', '?')} wg={r.get('Workgroup_Size', '?')}")
        tot_busy += e - s
        prev_end = e
    print(f"  == {len(seq)} kernels, span {(prev_end - t0) / 1e3:.1f} us, busy {tot_busy / 1e3:.1f} us")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 8, chunk=int(os.environ.get("CHUNK", "0")))
    else:
        show(sys.argv[2])
