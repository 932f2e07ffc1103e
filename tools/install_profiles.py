#!/usr/bin/env python3
"""Copies one run of tools/collect_profiles.sh (gpurun_out/r06) into profiles/, keeping the explanatory headers of the files that have one and the medians of
the earlier fence-cost runs.  usage: python tools/install_profiles.py [gpurun_out/r06]"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r06")
dst = os.path.join(ROOT, "profiles")


def header_of(path):
    if not os.path.exists(path):
        return ""
    out = []
    for line in open(path):
        if not line.startswith("#"):
            break
        out.append(line)
    return "".join(out)


keep_header = {"r06_seam_first_calls.txt", "r06_stateless_pcie_inclusive.txt"}
for f in sorted(glob.glob(os.path.join(src, "r06_*"))):
    name = os.path.basename(f)
    if name in ("r06_bench.err", "r06_pmc_traffic.log"):
        continue
    target = os.path.join(dst, name)
    if name == "r06_cold.jsonl":                       # -> the readable table of r06_seam_cold_process.txt
        target = os.path.join(dst, "r06_seam_cold_process.txt")
        hdr = header_of(target)
        rows = []
        for line in open(f):
            d = json.loads(line)
            rows.append(f"{d['warm']} {d['host_work_before_the_first_call_ms']} {[round(x, 2) for x in d['call_ms']]} {round(d['sum_ms'], 1)} {d['results_identical']}")
        open(target, "w").write(hdr + "\n".join(rows) + "\n")
        continue
    if name == "r06_fence_cost.txt":                   # the earlier runs' medians stay in front of the final build's samples
        old = open(target).read() if os.path.exists(target) else ""
        cut = old.find("# final build:")
        head = old[:cut] if cut >= 0 else ""
        open(target, "w").write(head + "# final build:\n" + open(f).read())
        continue
    if name in keep_header:
        hdr = header_of(target)
        body = "".join(l for l in open(f) if not (l.startswith("#") and name == "r06_stateless_pcie_inclusive.txt"))
        new_hdr = header_of(f) if name == "r06_seam_first_calls.txt" and not hdr else ""
        open(target, "w").write((hdr or new_hdr) + ("".join(l for l in open(f) if not l.startswith("#")) if name == "r06_seam_first_calls.txt" else body))
        continue
    shutil.copyfile(f, target)
print("installed", len(glob.glob(os.path.join(src, "r06_*"))), "files; soak:", open(os.path.join(dst, "r06_soak.txt")).readline().strip())
