#!/bin/bash
# On the MI355X box: the 20-minute soak and 6x10^5 stressed steps of the shipped sum-check hand-over on the build in the tree (profiles/r06_soak_long.txt, r06_sc_stress_long.txt).
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/r06g; mkdir -p $out
sha=$(cd $root && python -c "from reef_amd import _ffi; print(_ffi.library_sources_sha16())")
python $root/tools/soak.py ${LONG_SOAK_SECONDS:-1200} 6 > $out/r06_soak_long.txt 2>&1
(echo "# reef_amd/_lib/sc_stress <ell> <steps> load, REEF_SC_FENCE=2 (the shipped hand-over since round 6), 100000 steps per grid; library sources $sha"
 run() { env REEF_SC_FENCE=2 $2 $root/reef_amd/_lib/sc_stress $1 $3 load; }
 n=${LONG_STRESS_STEPS:-100000}
 run 12 "REEF_SC_BLOCKS=2 REEF_SC_ITEMS=1 REEF_SC_SPLIT_MAX=0" $n
 run 18 "REEF_SC_BLOCKS=3 REEF_SC_ITEMS=1 REEF_SC_SPLIT_MAX=0" $n
 run 16 "REEF_SC_BLOCKS=16 REEF_SC_ITEMS=1 REEF_SC_SPLIT_MAX=0" $n
 run 18 "REEF_SC_BLOCKS=64 REEF_SC_ITEMS=1 REEF_SC_SPLIT_MAX=0" $n
 run 20 "REEF_SC_BLOCKS=2048 REEF_SC_ITEMS=1 REEF_SC_SPLIT_MAX=0 REEF_SC_ONE_LAUNCH_MAX=8192" $((n / 2))
 run 17 "" $n
 run 18 "REEF_SC_SPLIT_BLOCKS=1024 REEF_SC_SPLIT_MAX=65536" $((n / 2))
) > $out/r06_sc_stress_long.txt 2>&1
head -1 $out/r06_soak_long.txt; grep -c '"mismatched_steps": 0' $out/r06_sc_stress_long.txt
