#!/usr/bin/env python3
"""HBM traffic per kernel launch from rocprofv3 PMC passes (run on the MI355X box).

Runs `bench.py` twice under `rocprofv3 --pmc <counter> --kernel-trace` (FETCH_SIZE, then WRITE_SIZE:
separate passes, never combined with any other trace domain), condenses the counter_collection CSVs
to one line per dispatch and writes profiles-ready files:

    <out>/rNN_pmc_FETCH_SIZE_bench_counters.csv, <out>/rNN_pmc_WRITE_SIZE_bench_counters.csv
    <out>/rNN_pmc_traffic.json      average bytes per launch per kernel

Units: both counters are reported in KiB.  Correction factors (see the _comment in the JSON): on
this part FETCH_SIZE counts a coalesced stream at half its size and a 64-byte-granular gather at
full size, calibrated on this build's own kernels; WRITE_SIZE needs none.

Usage: python tools/pmc_traffic.py <outdir> [bench.py args ...]
"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RND = os.environ.get("REEF_ROUND", "r04")      # file prefix: profiles are named per round
outdir = os.path.abspath(sys.argv[1])
bench_args = sys.argv[2:] or ["--steps", "4", "--warmup", "1", "--msms-per-step", "1", "--no-cpu-baseline", "--no-replay", "--streams", "1"]
os.makedirs(outdir, exist_ok=True)

GATHER_KERNELS = ("k_accum0",)          # 64-B point gathers: FETCH_SIZE is exact (factor 1.0)


def short(name: str) -> str:
    name = name.replace("void ", "").replace("reef::", "")
    return name.split("(")[0]


per = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d = os.path.join("/tmp", f"pmc_{counter}")
    subprocess.run(["rm", "-rf", d])
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
           os.path.join(ROOT, "bench.py")] + bench_args
    subprocess.run(cmd, cwd="/tmp", env=env, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
    vals = defaultdict(float)
    names = {}
    for r in csv.DictReader(open(cc)):
        if r["Counter_Name"] == counter:
            vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
    with open(os.path.join(outdir, f"{RND}_pmc_{counter}_bench_counters.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Dispatch_Id", "Counter", "Value_KiB", "DurationNs"])
        for did in sorted(vals, key=int):
            w.writerow([names[did], did, counter, f"{vals[did]:.6f}", dur.get(did, "")])
    agg = defaultdict(list)
    for did, v in vals.items():
        agg[names[did]].append(v)
    per[counter] = {k: sum(v) / len(v) for k, v in agg.items()}

line = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + bench_args, capture_output=True, text=True, timeout=600).stdout
cfg = json.loads(line.strip().splitlines()[-1])["config"]
kernels = {}
for k in sorted(per["FETCH_SIZE"]):
    f, w = per["FETCH_SIZE"][k], per["WRITE_SIZE"].get(k, 0.0)
    factor = 1.0 if k.startswith(GATHER_KERNELS) else 2.0
    kernels[k] = {"FETCH_SIZE": f, "WRITE_SIZE": w, "fetch_factor": factor, "hbm_bytes_per_launch": (f * factor + w) * 1024}
json.dump({
    "_comment": "HBM traffic per launch from rocprofv3 --pmc passes (separate runs: FETCH_SIZE, WRITE_SIZE; --kernel-trace only). "
                "Units KiB. Calibration on this build's own access patterns: the ubench 64-B gather probe (4 x 16-B loads per lane, the "
                "k_accum0 point fetch) reads 2097152 KiB and FETCH_SIZE reports 2142538 (x1.02 -> factor 1.0); k_recode's coalesced 32-B "
                "stream of 32768 KiB reports 16415 (factor 2.0, the gfx950 half-count of MI355X_MICROARCH.md); WRITE_SIZE of k_recode "
                "reports 65536 KiB for 65536 KiB written (factor 1.0).",
    "command": "python tools/pmc_traffic.py <out> " + " ".join(bench_args),
    "kernel_sources_sha16": __import__("reef_amd._ffi", fromlist=["x"]).kernel_sources_sha16(),
    "config": {"curve": "pallas", "logn": cfg["points_per_gpu"].bit_length() - 1, "window_bits": cfg["window_bits"],
               "bucket_groups": cfg["bucket_groups"]},
    "kernels": kernels,
}, open(os.path.join(outdir, f"{RND}_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in kernels.items()}))
