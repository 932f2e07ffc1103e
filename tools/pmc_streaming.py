#!/usr/bin/env python3
"""Rows N2 / N3, the HBM-bound kernels: HBM bytes per launch from the PMC counters (FETCH_SIZE and WRITE_SIZE in separate
rocprofv3 passes, corrected as tools/pmc_traffic.py does) next to the ALGORITHMIC bytes of the largest launches and their
duration, i.e. the fraction of the 8 TB/s peak each kernel reaches and how much of its traffic is waste.

    python tools/pmc_streaming.py <out.json>       (on the MI355X box; sum-check at ell = 26, bound rows at 2^25 x 32 B)
"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK = 8.0e12
CMDS = {"sumcheck": [os.path.join(ROOT, "tools", "time_sumcheck.py"), "26"], "mle": [os.path.join(ROOT, "tools", "time_mle.py")]}
# algorithmic bytes of the LARGEST launch of each kernel in those commands (entries of 32 B)
N26, N25 = 1 << 26, 1 << 25
ALGO = {
    "k_sc_coeffs<1>": ("round 1 at ell = 26: reads both tables once", 2 * N26 * 32),
    "k_sc_fold<1>": ("fold with pow = 2^25: reads both tables, writes half of each", 2 * N26 * 32 + 2 * N25 * 32),
    "k_sc_fold_coeffs<1>": ("fused fold + next coefficients with pow = 2^25: reads both tables, writes half of each", 2 * N26 * 32 + 2 * N25 * 32),
    "k_sc_r1_coeffs<1>": ("round 1 at ell = 26 with EQ rank-one: reads T once (EQ is two factor tables of 2^13 entries)", N26 * 32),
    "k_sc_r1_fold_coeffs<1, false>": ("fused fold + next coefficients with pow = 2^25, EQ rank-one: reads T, writes half of it", N26 * 32 + N25 * 32),
    # the hybrid-shaped table of tools/time_sumcheck.py at ell = 26: 8192 rows of 8192 entries; second half = 3584 rows of symbols + 512 zero rows
    "k_sc_r1s_rows<1, false>": ("round-1 sums over the 3584 symbol rows of the hybrid-shaped table: 4-byte entries read once", 3584 * 8192 * 4),
    "k_sc_r1_fold_coeffs_cs<1, false>": ("first fold of the hybrid-shaped table (2047 of 2048 row quadruples constant-or-small): 4-byte reads of the "
                                         "symbol rows, the folded table written as field elements", 3584 * 8192 * 4 + 2 * 2047 * 8192 * 32),
    "k_sc_eq_table<1>": ("eq table of 2^26 entries: written once (its two factor tables are cache-resident)", N26 * 32),
    "k_mle_bound<1, 32>": ("bound rows of a 2^25 x 32 B table: read once", N25 * 32),
}


def short(name):
    return name.replace("void ", "").replace("reef::", "").split("(")[0]


per = defaultdict(lambda: defaultdict(dict))      # kernel -> dispatch key -> {counter: value, dur}
for tag, cmd in CMDS.items():
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = f"/tmp/pmc_stream_{tag}_{counter}"
        subprocess.run(["rm", "-rf", d])
        p = subprocess.run(["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable] + cmd, cwd="/tmp",
                           env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=1200)
        cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if p.returncode != 0 or not cc or not kt:
            print(tag, counter, "failed:", p.stderr[-300:], file=sys.stderr)
            continue
        dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt[0]))}
        vals, names = defaultdict(float), {}
        for r in csv.DictReader(open(cc[0])):
            if r["Counter_Name"] == counter:
                vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
                names[r["Dispatch_Id"]] = short(r["Kernel_Name"])
        # dispatch ids differ between the two passes: key the launches of a kernel by their order
        order = defaultdict(int)
        for did in sorted(vals, key=int):
            k = names[did]
            per[k][order[k]][counter] = vals[did] * 1024.0     # KiB -> bytes
            per[k][order[k]].setdefault("dur_ns", []).append(dur.get(did, 0))
            order[k] += 1
out = {"_comment": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only); FETCH_SIZE of a coalesced stream "
                   "is reported at half its size on gfx950 (factor 2.0, calibrated in profiles/r03_pmc_traffic.json); durations are those of the counter "
                   "passes (counters stretch a launch a little: the timing files hold the undisturbed figures)",
       "hbm_peak_bytes_per_s": HBM_PEAK, "kernels": {}}
for k, (what, algo) in ALGO.items():
    if k not in per:
        continue
    # the largest launch = the one with the largest fetch
    best = max(per[k].values(), key=lambda v: v.get("FETCH_SIZE", 0.0))
    fetch = best.get("FETCH_SIZE", 0.0) * 2.0
    write = best.get("WRITE_SIZE", 0.0)
    dur = min(x for x in best.get("dur_ns", [0]) if x) if any(best.get("dur_ns", [0])) else 0
    row = {"launch": what, "algorithmic_bytes": algo, "hbm_read_bytes": fetch, "hbm_write_bytes": write, "hbm_bytes": fetch + write,
           "traffic_over_algorithmic": (fetch + write) / algo if algo else None, "duration_us": dur / 1e3}
    if dur:
        row["achieved_algorithmic_GBps"] = algo / (dur * 1e-9) / 1e9
        row["frac_of_hbm_peak"] = algo / (dur * 1e-9) / HBM_PEAK
    out["kernels"][k] = row
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
