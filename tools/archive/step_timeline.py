#!/usr/bin/env python3
"""Kernel timeline of the LAST sum-check step in a rocprofv3 --kernel-trace CSV (from its k_sc_eq_factors on).  usage: step_timeline.py <csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_sc_eq_factors" in r["Kernel_Name"]][-1]
t0 = int(rows[idx]["Start_Timestamp"]); prev = t0
for r in rows[idx:idx + 80]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void reef::", "").split("(")[0][:46]
    print("%-46s start %8.1f dur %7.1f gap %6.1f grid %sx%s wg %s" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"]))
    prev = e
