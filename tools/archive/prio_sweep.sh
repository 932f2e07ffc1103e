#!/bin/bash
# Tail kernels (and optionally the sort) of every MSM on a HIGH-priority companion stream, the accumulation on the ctx's stream or a
# LOW-priority one.  usage: tools/prio_sweep.sh
root=${GRAFT_REPO_ROOT:-.}
for mode in "0 0 0" "1 0 0" "1 1 0" "1 0 1" "1 1 1"; do set -- $mode
  REEF_MSM_PRIO_TAIL=$1 REEF_MSM_PRIO_SORT=$2 REEF_MSM_PRIO_ACC_LOW=$3 python $root/bench.py --no-cpu-baseline --no-replay 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['single_stream']
print('tails on a high-priority stream: $1  sort there too: $2  accumulation on a low-priority stream: $3   %.3f ms/MSM  k_accum0 %.3f  alone %.3f (%.3f)  issue.frac %.3f  %s' % (d['config']['ms_per_msm'], r['kernel_ms'], s['msm_ms'], s['kernel_ms'], r['issue']['frac'], d['config']['check']))"
done
