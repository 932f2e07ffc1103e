#!/bin/bash
# Group-operation latency probe under rocprofv3 (on the GPU box): prints the kernel durations of tools/coop_probe.py.
d=/tmp/pr_probe; rm -rf $d
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $d -- python $GRAFT_REPO_ROOT/tools/coop_probe.py > $d.log 2>&1)
f=$(ls $d/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_test_ec" in r["Kernel_Name"]]
for r in rows[-4:]:
    print(r["Kernel_Name"][:40], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
