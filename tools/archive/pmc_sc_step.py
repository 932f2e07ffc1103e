#!/usr/bin/env python3
"""VALU instructions of the kernels of one sum-check folding step (run on the MI355X box): is a mid-sized round bound by bandwidth or by
integer issue?  One rocprofv3 --pmc pass (never combined with other trace domains) over tools/step_breakdown.py; prints, per kernel
launch of the LAST step, wave-instructions, the time they need at the chip's measured issue rate (3.27e13 lane-instructions/s =
5.1e11 wave-instructions/s, profiles/r01_ubench_instruction_rates.txt) and the launch's grid.  Usage: python tools/pmc_sc_step.py <ell> <out.json>"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ell = sys.argv[1] if len(sys.argv) > 1 else "21"
ISSUE_WAVE_INSTS_PER_S = 3.27e13 / 64
d = "/tmp/pmc_sc"
subprocess.run(["rm", "-rf", d])
cmd = ["rocprofv3", "--pmc", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
       os.path.join(ROOT, "tools", "step_breakdown.py"), ell]
p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=900)
cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
if p.returncode != 0 or not cc:
    sys.exit("rocprofv3 failed: " + p.stderr[-400:])
disp = {}
for r in csv.DictReader(open(cc[0])):
    k = int(r["Dispatch_Id"])
    e = disp.setdefault(k, {"kernel": r["Kernel_Name"].replace("void ", "").replace("reef::", "").split("(")[0], "grid": int(r["Grid_Size"])})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
rows = [disp[k] for k in sorted(disp)]
# the last step: from the last k_sc_eq_factors on
start = max(i for i, r in enumerate(rows) if r["kernel"].startswith("k_sc_eq_factors"))
out = []
for r in rows[start:]:
    v = r.get("SQ_INSTS_VALU", 0.0)
    out.append({"kernel": r["kernel"], "grid_threads": r["grid"], "valu_wave_insts": v, "issue_floor_us": v / ISSUE_WAVE_INSTS_PER_S * 1e6,
                "vmem_rd_wave_insts": r.get("SQ_INSTS_VMEM_RD", 0.0), "vmem_wr_wave_insts": r.get("SQ_INSTS_VMEM_WR", 0.0)})
if len(sys.argv) > 2:
    json.dump({"ell": int(ell), "issue_rate_wave_insts_per_s": ISSUE_WAVE_INSTS_PER_S, "kernels_of_the_last_step": out}, open(sys.argv[2], "w"), indent=1)
for o in out:
    if o["issue_floor_us"] >= 3:
        print(f'{o["kernel"]:42s} grid {o["grid_threads"]:9d}  VALU {o["valu_wave_insts"]:12.0f} wave-insts = {o["issue_floor_us"]:7.1f} us at the issue rate')
