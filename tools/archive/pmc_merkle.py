#!/usr/bin/env python3
"""Row N4: VALU instructions per Poseidon hash of the Merkle kernels (sparse partial rounds, the shipped form) from the PMC
counters, and the issue-rate statement that follows (run on the MI355X box).

One rocprofv3 --pmc pass (counters only, with --kernel-trace for the durations) over tools/time_merkle.py at 2^24 symbols; the
leaf kernel hashes 2^23 nodes in one dispatch.  Usage: python tools/pmc_merkle.py <out.json>
"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LOGN = 24
LANE_INSTR_PER_S = 34.1e12      # v_mad_u64_u32 chip-wide (profiles/r01_ubench_instruction_rates.txt); the product is MAD-dominated
res = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for group in (["SQ_INSTS_VALU", "SQ_WAVES"], ["GRBM_GUI_ACTIVE"]):
    d = "/tmp/pmc_merkle"
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--pmc"] + group + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
           os.path.join(ROOT, "tools", "time_merkle.py"), str(LOGN)]
    p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=900)
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if p.returncode != 0 or not cc:
        print("group", group, "failed:", p.stderr[-300:], file=sys.stderr)
        continue
    per = defaultdict(lambda: defaultdict(float))
    grid = {}
    for r in csv.DictReader(open(cc[0])):
        name = r["Kernel_Name"].replace("void ", "").replace("reef::", "").split("(")[0]
        per[(name, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        grid[(name, r["Dispatch_Id"])] = int(r.get("Grid_Size", 0) or 0)
    for (name, did), cs in per.items():
        for c, v in cs.items():
            res[(name, grid[(name, did)])][c].append(v)
    for kt in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(kt)):
            name = r["Kernel_Name"].replace("void ", "").replace("reef::", "").split("(")[0]
            dur[(name, int(r.get("Grid_Size", 0) or 0))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = {"document_symbols": 1 << LOGN, "kernels": []}
for (name, grid), cs in sorted(res.items(), key=lambda kv: -kv[0][1]):
    if not name.startswith("k_pos"):
        continue
    row = {"kernel": name, "grid_threads": grid, **{c: sum(v) / len(v) for c, v in cs.items()}}
    if (name, grid) in dur:
        row["duration_us"] = min(dur[(name, grid)])          # counters stretch a launch: the trace of the same pass, shortest launch
    out["kernels"].append(row)
big = [k for k in out["kernels"] if k["kernel"].startswith("k_pos_leaves") and "SQ_INSTS_VALU" in k]
if big:
    k = big[0]
    hashes = (1 << LOGN) // 2
    per_hash = k["SQ_INSTS_VALU"] * 64.0 / max(k["grid_threads"], 1) if k["grid_threads"] else None   # wave instructions -> per lane = per hash
    out["valu_instructions_per_hash"] = per_hash
    out["note"] = ("SQ_INSTS_VALU counts wave instructions; one thread hashes one node, so instructions per hash = wave instructions * 64 / threads. "
                   "Issue-rate statement: hashes/s * instructions per hash against the chip's %.1f T lane-instructions/s" % (LANE_INSTR_PER_S / 1e12))
    out["leaf_kernel_hashes"] = hashes
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
