#!/bin/bash
# On the MI355X box: what profiles/ holds about the FINAL build of a round (the release build, reef_amd/_lib/libreef_msm.so).
# usage: tools/collect_profiles.sh OUTDIR [rNN]        (~25 GPU-minutes; SOAK_SECONDS shortens the soak)
# Order: PMC traffic first (bench.py reads roofline.traffic from profiles/ and refuses another build's), then the bench line and its rocprofv3 twin, the
# seam, the replays, the HBM-bound rows, the stress run and -- last, because it vouches for everything before it -- the soak.
out=$(realpath -m $1); R=${2:-r06}; export REEF_ROUND=$R
mkdir -p $out; export TMPDIR=/tmp; root=${GRAFT_REPO_ROOT:-$(pwd)}
sha=$(cd $root && python -c "from reef_amd import _ffi; print(_ffi.library_sources_sha16())")
python $root/tools/pmc_traffic.py $out > $out/${R}_pmc_traffic.log 2>&1
mkdir -p $root/profiles; cp $out/${R}_pmc_traffic.json $root/profiles/ 2>/dev/null
python $root/tools/pmc_valu.py $out/${R}_pmc_valu_issue.json > /dev/null 2>&1
python $root/bench.py --steps 20 --warmup 5 > $out/${R}_bench.json 2> $out/${R}_bench.err
(cd /tmp && rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-replay > $out/${R}_bench_under_rocprof.json 2>/dev/null; cp /tmp/ks/*/*kernel_stats.csv $out/${R}_kernel_stats.csv)
# the seam: the first calls on a fresh key, then the steady state under 1 / 4 / 8 callers
(echo "# reef_amd/_lib/seam_bench first=1 (release build, library sources $sha): the first eight calls on a key the process has never seen, ms; gap_us = host time between the calls"
 $root/reef_amd/_lib/seam_bench first=1 sizes=4096,16384,27790,65536,131072,1048576 gap_us=0,1000) > $out/${R}_seam_first_calls.txt 2>/dev/null
$root/reef_amd/_lib/seam_bench > $out/${R}_stateless_concurrent.txt 2>&1
python $root/tools/time_stateless.py > $out/${R}_stateless_pcie_inclusive.txt 2>&1
(python $root/tools/time_plain_key.py; echo "# REEF_MSM_HOST_COMBINE=0 (the window combine on the device, rounds 1-5):"; REEF_MSM_HOST_COMBINE=0 python $root/tools/time_plain_key.py 12 15 20) 2>&1 | grep -v amdgpu.ids > $out/${R}_plain_key_device_results.txt
python $root/tools/sweep_plans.py 12 14 15 16 17 18 20 2>&1 | grep "##" > $out/${R}_latency_sweep.txt
python $root/tools/time_setup.py $out/${R}_setup_timing.json > /dev/null 2>&1
# Reef's own MSM sequence, replayed and checked (set-up on the clock since round 6), and the same under rocprofv3: HBM GB/s per kernel (BASELINE configs[2])
for c in cfg1 cfg3 cfg4 cfg5; do $root/reef_amd/_lib/reef_replay $c nofold; done > $out/${R}_replay_prove_msm.jsonl 2>/dev/null
$root/reef_amd/_lib/reef_replay cfg3 >> $out/${R}_replay_prove_msm.jsonl 2>/dev/null
for c in cfg3 cfg4; do $root/reef_amd/_lib/reef_replay $c nofold tables; done >> $out/${R}_replay_prove_msm.jsonl 2>/dev/null
for c in cfg3 cfg4 cfg5; do python $root/tools/pmc_replay.py $out $c > /dev/null 2>&1; done
# the multi-GPU split as one process sees it (ordinals repeat on a one-GPU box: labelled), phases itemised
(python $root/bench.py --gpus 2 --single-process --steps 3 --warmup 1 --msms-per-step 12; python $root/bench.py --gpus 8 --single-process --steps 3 --warmup 1 --msms-per-step 12
 python $root/bench.py --gpus 8 --single-process --group-exchange rccl --steps 3 --warmup 1 --msms-per-step 12) 2>/dev/null | grep "^{" > $out/${R}_bench_single_process.jsonl
(for m in 3 8; do $root/reef_amd/_lib/reef_replay cfg4 nofold devices=$m; done; $root/reef_amd/_lib/reef_replay cfg5 nofold devices=8) > $out/${R}_replay_devices.jsonl 2>/dev/null
python $root/tools/time_group.py > $out/${R}_group_timing.txt 2>&1
# the HBM-bound rows
python $root/tools/pmc_streaming.py $out/${R}_pmc_streaming.json > /dev/null 2>&1
(echo "# experiment build (the dense / unstructured A-B rows need its switches)"; REEF_MSM_LIB=$root/reef_amd/_lib/libreef_msm_exp.so python $root/tools/time_sumcheck.py 21 26) > $out/${R}_sumcheck_timing.txt 2>&1
python $root/tools/time_mle.py > $out/${R}_mle_timing.txt 2>&1
python $root/tools/time_merkle.py 16 20 24 26 27 > $out/${R}_merkle_timing.txt 2>&1
python $root/tools/time_keygen.py $out/${R}_keygen_timing.json > /dev/null 2>&1
bash $root/tools/fence_cost.sh 5 > $out/${R}_fence_cost.txt 2>&1
# the one-launch sum-check rounds under load: the shipped hand-over (REEF_SC_FENCE=2) ten times the GPU suite's count, the other two forms at its count
(echo "# reef_amd/_lib/sc_stress <ell> <steps> load (experiment build: the grids are forced with its switches): one folding step repeated under k_accum0 + streaming load, every coefficient triple against the two-launch form; library sources $sha"
 run() { env REEF_SC_FENCE=$1 $3 $root/reef_amd/_lib/sc_stress $2 $4 load; }
 for f in 2 0 1; do n=$([ $f = 2 ] && echo 20000 || echo 4000)
   run $f 12 "REEF_SC_BLOCKS=2 REEF_SC_ITEMS=1 REEF_SC_SPLIT_MAX=0" $n
   run $f 18 "REEF_SC_BLOCKS=3 REEF_SC_ITEMS=1 REEF_SC_SPLIT_MAX=0" $n
   run $f 16 "REEF_SC_BLOCKS=16 REEF_SC_ITEMS=1 REEF_SC_SPLIT_MAX=0" $n
   run $f 18 "REEF_SC_BLOCKS=64 REEF_SC_ITEMS=1 REEF_SC_SPLIT_MAX=0" $n
   run $f 20 "REEF_SC_BLOCKS=2048 REEF_SC_ITEMS=1 REEF_SC_SPLIT_MAX=0 REEF_SC_ONE_LAUNCH_MAX=8192" $((n / 2))
   run $f 17 "" $n
   run $f 18 "REEF_SC_SPLIT_BLOCKS=1024 REEF_SC_SPLIT_MAX=65536" $((n / 2))
   run $f 16 "REEF_SC_RANK1_MIN_POW=1" $((n / 2))
 done) > $out/${R}_sc_stress.txt 2>&1
# the soak LAST, on the build everything above was measured on: it records the fingerprint of the library's sources, and
# tests/test_profiles_fresh.py refuses a soak of other sources
python $root/tools/soak.py ${SOAK_SECONDS:-420} 6 > $out/${R}_soak.txt 2>&1
