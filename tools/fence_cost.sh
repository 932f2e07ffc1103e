#!/bin/bash
# what does the release-ordered ticket (REEF_SC_FENCE=2, the form the HIP memory model blesses) cost a folding step's sum-check
# against the fence-free hand-over (REEF_SC_FENCE=0)?  reef_replay's sumcheck_ms_per_step (median of three steps inside the harness) for cfg3 and cfg4,
# `reps` alternating repetitions on one box; prints every sample and the medians.
# usage: tools/fence_cost.sh [reps=5]
root=${GRAFT_REPO_ROOT:-.}; reps=${1:-5}
for cfg in cfg3 cfg4; do
  for r in $(seq $reps); do for f in 0 2 1; do
    v=$(REEF_SC_FENCE=$f $root/reef_amd/_lib/reef_replay $cfg nofold 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['sumcheck_ms_per_step'], d['sumcheck_table_log'])")
    echo "$cfg REEF_SC_FENCE=$f rep $r: sumcheck_ms_per_step table_log = $v"
  done; done
done | tee /tmp/fence_samples.txt
python3 - <<'PY'
import re, statistics
s = {}
for line in open('/tmp/fence_samples.txt'):
    m = re.match(r"(cfg\d) REEF_SC_FENCE=(\d) rep \d+: .* = ([\d.]+) (\d+)", line)
    if m: s.setdefault((m.group(1), m.group(2)), []).append(float(m.group(3)))
for cfg in ('cfg3', 'cfg4'):
    if (cfg, '0') in s:
        m0, m2, m1 = (statistics.median(s[(cfg, f)]) for f in ('0', '2', '1'))
        print(f"## {cfg}: medians  fence-free {m0:.4f} ms   acq_rel ticket (2) {m2:.4f} ms ({100*(m2/m0-1):+.2f} %)   __threadfence (1) {m1:.4f} ms ({100*(m1/m0-1):+.2f} %)")
PY
