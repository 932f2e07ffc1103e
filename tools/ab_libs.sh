#!/bin/bash
# A/B of library builds and switches on ONE box, alternating, `passes` times: boxes differ by 5-8 %, runs on one box by ~1 %.
# usage: tools/ab_libs.sh passes "label|ENV=.. ENV=..|lib.so" ...      (lib.so relative to reef_amd/_lib/variants/, empty = the tree's build)
root=${GRAFT_REPO_ROOT:-.}; passes=$1; shift
for p in $(seq $passes); do for spec in "$@"; do
  IFS='|' read -r label envs lib <<< "$spec"
  libenv=""; [ -n "$lib" ] && libenv="REEF_MSM_LIB=$root/reef_amd/_lib/variants/$lib"
  env $envs $libenv python $root/bench.py --no-cpu-baseline --no-replay 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['single_stream']
print('%-44s %.4f ms/MSM  k_accum0 %.3f  alone %.3f (%.3f)  issue.frac %.3f  %s' % ('$label', d['config']['ms_per_msm'], r['kernel_ms'], s['msm_ms'], s['kernel_ms'], r['issue']['frac'], d['config']['check']))"
done; done
