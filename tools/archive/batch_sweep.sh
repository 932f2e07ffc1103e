#!/bin/bash
# bench.py with b MSMs per step issued as ONE batched call (reef_msm_rows, rows = b) against b = 1 with several streams.
# usage: tools/batch_sweep.sh [OUTFILE]
out=${1:-gpurun_out/batch_sweep.txt}; mkdir -p $(dirname $out); root=$(dirname $(dirname $(realpath $0)))
for cfg in "1 3" "1 4" "1 6" "2 1" "2 2" "3 1" "3 2" "4 1" "4 2" "6 1" "6 2"; do set -- $cfg
  python $root/bench.py --batch $1 --streams $2 --msms-per-step 1 --steps 40 --no-cpu-baseline --no-replay 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print('MSMs per call',c['msms_per_step'],'streams',c['streams'],'value %.1f M pairs/s'%(d['value']/1e6),'ms per call %.3f'%d['ms_per_step'],'per MSM %.3f'%(d['ms_per_step']/c['msms_per_step']),'check',c['check'],'issue %.3f'%d['roofline']['issue']['frac'])"
done > $out 2>&1
cat $out
