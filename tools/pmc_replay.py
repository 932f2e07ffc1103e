#!/usr/bin/env python3
"""BASELINE configs[2] ("1 MiB ASCII doc, -r '.*password.*' --prove, 1 MI355X, rocprof HBM GB/s reported"): the replay of that run's GPU
work (reef_amd/_lib/reef_replay cfg3 nofold: every MSM, IPA round, sum-check step, row binding and key derivation of the run through the
C ABI, commitments checked) under rocprofv3 -- one --kernel-trace pass for the durations, then one --pmc pass each for FETCH_SIZE and
WRITE_SIZE (separate runs, no other trace domain), condensed to one row per kernel: launches, time, HBM bytes per launch (the guide's
corrections as in tools/pmc_traffic.py: KiB units; FETCH_SIZE counts a coalesced stream at half its size on gfx950 -> x2, the 64-byte
gathers of k_accum0 at full size -> x1), achieved GB/s and the fraction of the 8 TB/s peak.

    python tools/pmc_replay.py <outdir> [cfg3|cfg4|...]     (on the MI355X box)  ->  <outdir>/rNN_<cfg>_prove_hbm.json / .txt
"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = os.environ.get("REEF_ROUND", "r05")
outdir = os.path.abspath(sys.argv[1])
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
os.makedirs(outdir, exist_ok=True)
exe = os.path.join(ROOT, "reef_amd", "_lib", "reef_replay")
PEAK = 8000.0
GATHER = ("k_accum0",)


def short(name):
    return name.replace("void ", "").replace("reef::", "").split("(")[0]


def run(extra, tag):
    d = f"/tmp/pr_{tag}"
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3"] + extra + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", exe, cfg, "nofold"]
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=900)
    if r.returncode != 0:
        raise SystemExit(f"{' '.join(cmd)} failed: {r.stderr[-800:]}")
    return d, r.stdout


d, out = run([], "trace")
line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
dur = defaultdict(list)
for r in csv.DictReader(open(kt)):
    dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
per = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    d, _ = run(["--pmc", counter], counter)
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    vals, names = defaultdict(float), {}
    for r in csv.DictReader(open(cc)):
        if r["Counter_Name"] == counter:
            vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
            names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
    agg = defaultdict(list)
    for did, v in vals.items():
        agg[names[did]].append(v)
    per[counter] = {k: sum(v) / len(v) for k, v in agg.items()}

rows = []
for k, ds in dur.items():
    f, w = per["FETCH_SIZE"].get(k, 0.0), per["WRITE_SIZE"].get(k, 0.0)
    factor = 1.0 if k.startswith(GATHER) else 2.0
    by = (f * factor + w) * 1024
    avg_ns = sum(ds) / len(ds)
    rows.append({"kernel": k, "launches": len(ds), "total_ms": sum(ds) / 1e6, "avg_us": avg_ns / 1e3, "hbm_bytes_per_launch": by, "fetch_factor": factor,
                 "achieved_GBps": by / avg_ns if avg_ns else 0.0, "frac_of_peak": by / avg_ns / PEAK if avg_ns else 0.0})
rows.sort(key=lambda r: -r["total_ms"])
total_ms = sum(r["total_ms"] for r in rows)
total_bytes = sum(r["hbm_bytes_per_launch"] * r["launches"] for r in rows)
doc = {"_comment": "rocprofv3 passes over `reef_replay " + cfg + " nofold` (the GPU work of one `reef --prove` of BASELINE's config replayed through the C ABI, setup included): "
                   "--kernel-trace for the durations; --pmc FETCH_SIZE and --pmc WRITE_SIZE in runs of their own (KiB; FETCH_SIZE x2 for coalesced streams on gfx950, x1 for "
                   "k_accum0's 64-byte gathers: calibration in profiles/r05_pmc_traffic.json).  achieved_GBps = corrected HBM bytes per launch / average launch duration.",
       "command": "python tools/pmc_replay.py <out> " + cfg, "replay_line": {k: line[k] for k in ("replay", "w1", "w2", "steps", "total_prove_msm_ms", "total_prove_gpu_ms", "sumcheck_ms_per_step") if k in line},
       "hbm_peak_GBps": PEAK, "kernel_time_ms": total_ms, "hbm_bytes_total": total_bytes, "whole_run_GBps_over_kernel_time": total_bytes / (total_ms * 1e6) if total_ms else 0.0,
       "kernels": rows}
json.dump(doc, open(os.path.join(outdir, f"{RND}_{cfg}_prove_hbm.json"), "w"), indent=1)
with open(os.path.join(outdir, f"{RND}_{cfg}_prove_hbm.txt"), "w") as f:
    f.write(f"# {doc['command']}: per-kernel HBM traffic and rate of the replayed --prove GPU work ({line.get('replay')}); peak {PEAK:.0f} GB/s\n")
    f.write(f"# kernel time {total_ms:.2f} ms, {total_bytes / 1e9:.3f} GB of HBM traffic: {doc['whole_run_GBps_over_kernel_time']:.0f} GB/s over the kernels' own time\n")
    f.write(f"{'kernel':60s} {'launches':>8s} {'total ms':>9s} {'avg us':>9s} {'MB/launch':>10s} {'GB/s':>8s} {'of peak':>8s}\n")
    for r in rows:
        f.write(f"{r['kernel'][:60]:60s} {r['launches']:8d} {r['total_ms']:9.3f} {r['avg_us']:9.1f} {r['hbm_bytes_per_launch'] / 1e6:10.3f} {r['achieved_GBps']:8.1f} {r['frac_of_peak']:8.4f}\n")
print(open(os.path.join(outdir, f"{RND}_{cfg}_prove_hbm.txt")).read()[:6000])
