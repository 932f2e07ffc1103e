#!/usr/bin/env python3
"""VALU issue counters of the bucket-accumulation kernel (run on the MI355X box).

One rocprofv3 --pmc pass per counter group (never combined with other trace domains) over
`bench.py --streams 1`; prints per-launch averages for k_accum0 and writes them as JSON.
Usage: python tools/pmc_valu.py <out.json>
"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GROUPS = [["SQ_INSTS_VALU", "SQ_INSTS_SALU"], ["SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"], ["SQ_WAVE_CYCLES", "SQ_WAVES"],
          ["GRBM_GUI_ACTIVE"], ["SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"], ["SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]]
res = defaultdict(dict)
for g in GROUPS:
    d = "/tmp/pmc_valu"
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--pmc"] + g + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
           os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--msms-per-step", "1", "--no-cpu-baseline", "--no-replay", "--streams", "1", "--no-check"]
    p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=600)
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if p.returncode != 0 or not cc:
        print("group", g, "failed:", p.stderr[-300:], file=sys.stderr)
        continue
    acc = defaultdict(lambda: defaultdict(list))
    per_dispatch = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(cc[0])):
        name = r["Kernel_Name"].replace("void ", "").replace("reef::", "").split("(")[0]
        per_dispatch[(name, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (name, _), cs in per_dispatch.items():
        for c, v in cs.items():
            acc[name][c].append(v)
    for name, cs in acc.items():
        for c, v in cs.items():
            res[name][c] = sum(v) / len(v)
out = {k: v for k, v in res.items() if k.startswith(("k_accum0", "k_accumN", "k_merge", "k_scatter", "k_count", "k_reduce", "k_fine"))}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out.get("k_accum0<0>", {}), indent=1))
