#!/usr/bin/env python3
"""gpurun_out/plain_trace_{12,20}.csv (tools/archive/plain_key_trace.sh: rocprofv3 --kernel-trace of tools/time_plain_key.py) -> the kernel timeline of ONE MSM
on a plain key, with the distance to the next MSM's first kernel: what lies between the last kernel and the 96-byte copy is the window combine on the host."""
import csv
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for logn in (12, 20):
    rows = list(csv.DictReader(open(os.path.join(root, "gpurun_out", f"plain_trace_{logn}.csv"))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void reef::k_recode<0>")]   # the plain key's recoder (the pre-shifted key's is k_recode_count)
    a, b = starts[10], starts[11]
    t0 = int(rows[a]["Start_Timestamp"])
    print(f"2^{logn}, plain key: one MSM of the timed loop (call + sync each); the next MSM's first kernel starts {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us after this one's")
    for r in rows[a:b]:
        print("   %8.1f +%7.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0][:60]))
