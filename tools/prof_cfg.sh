#!/bin/bash
# usage: tools/prof_cfg.sh OUTDIR LOGN C G [CHUNK] [both]   (on the GPU box) -> OUTDIR/trace_LOGN_C_G[_Lchunk][_nocoop].txt
out=$1; logn=$2; c=$3; g=$4; chunk=${5:-0}; both=$6
mkdir -p $out
export TMPDIR=/tmp
modes="1"; [ -n "$both" ] && modes="1 0"
for coop in $modes; do
  d=/tmp/prof_${logn}_${c}_${g}_${chunk}_${coop}; rm -rf $d
  (cd /tmp && CHUNK=$chunk REEF_MSM_COOP=$coop rocprofv3 --kernel-trace --output-format csv -d $d -- python $GRAFT_REPO_ROOT/tools/profile_one.py run $logn $c $g > $d.log 2>&1)
  f=$(ls $d/*/*kernel_trace.csv | head -1)
  tag=""; [ $chunk != 0 ] && tag="_L$chunk"; [ $coop = 0 ] && tag="${tag}_nocoop"
  (grep "^logn" $d.log; python $GRAFT_REPO_ROOT/tools/profile_one.py show $f) > $out/trace_${logn}_${c}_${g}${tag}.txt 2>&1
done
