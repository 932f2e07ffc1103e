#!/bin/bash
# VERDICT r5 item 3, priced before built: what would a larger Pippenger window be worth if its bucket reduction cost NOTHING?
# Experiment build, REEF_MSM_SKIP_REDUCE=1 leaves the three reduction kernels out (the result is garbage: --no-check), =2 the row / column
# sums only.  One box, alternating, `passes` times: bench.py's timed region with 3 MSMs of 2^20 points in flight (ms per MSM).
# usage: tools/free_reduction_bound.sh [passes=3]
root=${GRAFT_REPO_ROOT:-.}; passes=${1:-3}
export REEF_MSM_LIB=$root/reef_amd/_lib/libreef_msm_exp.so
run() {  # label, window, skip
  local extra=""; [ "$3" != 0 ] && extra="--no-check"
  REEF_MSM_SKIP_REDUCE=$3 python $root/bench.py --allow-experiment --steps 4 --warmup 2 --msms-per-step 48 --no-cpu-baseline --no-replay --window-bits $2 $extra 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['single_stream']
print('%-34s c=%-2d tables=%-2d  %.4f ms/MSM in flight   k_accum0 %.3f in flight   one MSM alone %.3f (k_accum0 %.3f)   %s' % ('$1', d['config']['window_bits'], d['config']['tables'], d['config']['ms_per_msm'], r['kernel_ms'], s['msm_ms'], s['kernel_ms'], d['config']['check']))"
}
for p in $(seq $passes); do
  run "shipped" 17 0
  run "reduction free" 17 1
  for c in 18 19 20; do
    run "as built" $c 0
    run "row/column sums free" $c 2
    run "reduction free" $c 1
  done
done
