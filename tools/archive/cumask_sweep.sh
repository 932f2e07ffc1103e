#!/bin/bash
# sort / merge / reduce kernels on CUs of their own (REEF_MSM_CUMASK_TAIL = CUs per XCD for them), the accumulation
# masked to the complement (REEF_MSM_CUMASK_ACC=0) or left unmasked (=1).  usage: tools/cumask_sweep.sh > profiles/r04_cumask_sweep.txt
root=${GRAFT_REPO_ROOT:-.}
echo "# python bench.py --no-cpu-baseline --no-replay; per line: CUs per XCD (of 32) for the tail kernels' masked stream, where the sort runs, accumulation mask, ms per 2^20-point MSM with 3 in flight, k_accum0 ms per launch in flight, one MSM alone (ms, of which k_accum0), check"
for k in 0 1 2 4 8; do for mode in "1 0" "0 0" "1 1" "0 1"; do set -- $mode; acc=$1; srt=$2
  if [ $k = 0 ] && [ "$mode" != "1 0" ]; then continue; fi
  REEF_MSM_CUMASK_TAIL=$k REEF_MSM_CUMASK_ACC=$acc REEF_MSM_CUMASK_SORT=$srt python $root/bench.py --no-cpu-baseline --no-replay 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['single_stream']
print('tail CUs/XCD $k  sort %s  acc %s  %.3f ms/MSM  k_accum0 %.3f  alone %.3f (%.3f)  issue.frac %.3f  %s' % ('with the tails' if $srt else 'unmasked      ', 'unmasked  ' if $acc else 'complement', d['config']['ms_per_msm'], r['kernel_ms'], s['msm_ms'], s['kernel_ms'], r['issue']['frac'], d['config']['check']))"
done; done
