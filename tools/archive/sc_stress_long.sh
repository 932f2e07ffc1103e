root=$GRAFT_REPO_ROOT; out=$root/gpurun_out/long; mkdir -p $out
sha=$(cd $root && python -c "from reef_amd import _ffi; print(_ffi.library_sources_sha16())")
(echo "# reef_amd/_lib/sc_stress <ell> <steps> load, REEF_SC_FENCE=0 (the shipped hand-over), ten times the steps of r05_sc_stress.txt; library sources $sha"
 run() { env REEF_SC_FENCE=0 $2 $root/reef_amd/_lib/sc_stress $1 $3 load; }
 n=${STRESS_STEPS:-200000}
 run 12 "REEF_SC_BLOCKS=2 REEF_SC_ITEMS=1 REEF_SC_SPLIT_MAX=0" $n
 run 18 "REEF_SC_BLOCKS=3 REEF_SC_ITEMS=1 REEF_SC_SPLIT_MAX=0" $n
 run 16 "REEF_SC_BLOCKS=16 REEF_SC_ITEMS=1 REEF_SC_SPLIT_MAX=0" $n
 run 18 "REEF_SC_BLOCKS=64 REEF_SC_ITEMS=1 REEF_SC_SPLIT_MAX=0" $n
 run 20 "REEF_SC_BLOCKS=2048 REEF_SC_ITEMS=1 REEF_SC_SPLIT_MAX=0 REEF_SC_ONE_LAUNCH_MAX=8192" $((n / 2))
 run 17 "" $n
 run 18 "REEF_SC_SPLIT_BLOCKS=1024 REEF_SC_SPLIT_MAX=65536" $((n / 2))
 run 16 "REEF_SC_RANK1_MIN_POW=1" $((n / 2))
) > $out/r05_sc_stress_long.txt 2>&1
grep -o '"mismatched_values": [0-9]*' $out/r05_sc_stress_long.txt | sort | uniq -c
