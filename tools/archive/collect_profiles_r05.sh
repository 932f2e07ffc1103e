#!/bin/bash
# On the MI355X box: everything profiles/ holds for a round.  usage: tools/collect_profiles.sh OUTDIR [rNN]
#        tools/collect_profiles.sh OUTDIR rNN final   only what vouches for the FINAL build: bench line (+ rocprofv3 twin, PMC traffic), stress run, soak
out=$(realpath -m $1); R=${2:-r05}; export REEF_ROUND=$R; MODE=${3:-all}
mkdir -p $out; export TMPDIR=/tmp; root=$GRAFT_REPO_ROOT
if [ "$MODE" = all ]; then
python $root/bench.py --steps 20 --warmup 5 > $out/${R}_bench.json 2> $out/${R}_bench.err
# per-kernel time of the same timed region: the legs that run other sizes through the same kernels after it (replay, CPU) are left out
(cd /tmp && rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-replay > $out/${R}_bench_under_rocprof.json 2>/dev/null; cp /tmp/ks/*/*kernel_stats.csv $out/${R}_kernel_stats.csv)
python $root/tools/pmc_traffic.py $out > $out/${R}_pmc_traffic.log 2>&1
python $root/tools/pmc_valu.py $out/${R}_pmc_valu_issue.json > /dev/null 2>&1
python $root/tools/sweep_plans.py 12 14 15 16 17 18 20 2>&1 | grep "##" > $out/${R}_latency_sweep.txt
python $root/tools/time_small.py > $out/${R}_small_msm_latency.txt 2>&1
(python $root/tools/archive/time_host_combine.py; REEF_MSM_HOST_COMBINE=0 python $root/tools/archive/time_host_combine.py) > $out/${R}_host_combine.txt 2>&1
for c in cfg1 cfg3 cfg4 cfg5; do $root/reef_amd/_lib/reef_replay $c nofold; done > $out/${R}_replay_prove_msm.jsonl 2>/dev/null
$root/reef_amd/_lib/reef_replay cfg3 >> $out/${R}_replay_prove_msm.jsonl 2>/dev/null
$root/reef_amd/_lib/reef_replay cfg4 >> $out/${R}_replay_prove_msm.jsonl 2>/dev/null
for c in cfg1 cfg3 cfg4 cfg5; do $root/reef_amd/_lib/reef_replay $c nofold tables; done >> $out/${R}_replay_prove_msm.jsonl 2>/dev/null
python $root/tools/time_ipa.py 14 15 16 > $out/${R}_ipa_round_timing.txt 2>&1
for cfg in "15 13 1" "16 15 1" "20 17 1"; do $root/tools/prof_cfg.sh /tmp/tr $cfg 0 both; done
cat /tmp/tr/trace_*.txt > $out/${R}_msm_kernel_timelines.txt
(cd /tmp && rm -rf /tmp/pp && rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -- python $root/tools/archive/coop_probe.py > /dev/null 2>&1; python - <<PY > $out/${R}_group_op_latency.txt
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/pp/*/*kernel_trace.csv')[0])))
rows = [r for r in rows if 'k_test_ec' in r['Kernel_Name']][-4:]
names = ['four-wave addition x256', 'four-wave doubling x64', 'one-wave addition x128', 'one-wave doubling x64']
cnt = [256, 64, 128, 64]
print('# tools/coop_probe.py under rocprofv3 --kernel-trace: chained group operations on ONE workgroup, us per operation')
for r, n, c in zip(rows, names, cnt):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print(f'{n:28s} {d:9.1f} us total  {d / c:6.2f} us per operation')
PY
)
python $root/tools/time_rows.py 1024 2048 131 > $out/${R}_rows_timing.txt 2>&1; python $root/tools/time_rows.py 4096 8192 7 >> $out/${R}_rows_timing.txt 2>&1
python $root/tools/time_sumcheck.py 21 26 > $out/${R}_sumcheck_timing.txt 2>&1
$root/reef_amd/_lib/sync_probe > $out/${R}_sync_probe.txt 2>&1
python $root/tools/time_mle.py > $out/${R}_mle_timing.txt 2>&1
(python $root/tools/time_merkle.py 16 20 24 26 27; echo "# REEF_POSEIDON_DENSE=1 (partial rounds in the defining dense form), same box:"; REEF_POSEIDON_DENSE=1 python $root/tools/time_merkle.py 20 24 26 | grep symbols) > $out/${R}_merkle_timing.txt 2>&1
python $root/tools/time_keygen.py $out/${R}_keygen_timing.json > /dev/null 2>&1
python $root/tools/time_setup.py $out/${R}_setup_timing.json > /dev/null 2>&1
(python $root/tools/archive/time_host_scalars.py 20; python $root/tools/archive/time_host_scalars.py 16; echo "# 2^16 points with byte tables:"; python $root/tools/archive/time_host_scalars.py 16 --tables) > $out/${R}_host_scalars.txt 2>/dev/null
# round 3 additions
$root/reef_amd/_lib/dp_probe > $out/${R}_dp_probe.txt 2>&1
python $root/tools/archive/cpu_scaling.py 18 > $out/${R}_cpu_scaling.txt 2>&1
python $root/tools/time_stateless.py > $out/${R}_stateless_pcie_inclusive.txt 2>&1
(REEF_MSM_GRAPH=0 python $root/tools/archive/time_graph.py; REEF_MSM_GRAPH=1 python $root/tools/archive/time_graph.py) > $out/${R}_graph_latency.txt 2>&1
python $root/tools/archive/pmc_merkle.py $out/${R}_pmc_merkle.json > /dev/null 2>&1
bash $root/tools/archive/batch_sweep.sh $out/${R}_batch_sweep.txt > /dev/null 2>&1
(echo "# python tools/time_merkle.py, one box, back to back: five lanes per hash on the levels of at most N nodes (REEF_POSEIDON_SPREAD_MAX=N; shipped: 16384), then a thread per node everywhere (REEF_POSEIDON_SPREAD=0)"
 for m in 2048 8192 16384 32768 65536; do echo "## N = $m"; REEF_POSEIDON_SPREAD_MAX=$m python $root/tools/time_merkle.py 10 14 16 18 20 22 | grep symbols; done
 echo "## a thread per node"; REEF_POSEIDON_SPREAD=0 python $root/tools/time_merkle.py 10 14 16 18 20 22 | grep symbols) > $out/${R}_merkle_spread.txt 2>&1
(for f in 1 0; do echo "REEF_MSM_FUSE_MERGE=$f"; REEF_MSM_FUSE_MERGE=$f python $root/tools/sweep_plans.py 12 14 15 16 17 2>&1 | grep "##" | grep "G=1"; done) > $out/${R}_fused_merge_ab.txt 2>&1
python $root/tools/pmc_streaming.py $out/${R}_pmc_streaming.json > /dev/null 2>&1
# round 4 additions
python $root/tools/time_stateless.py --threads 1,4,8 > $out/${R}_stateless_concurrent.txt 2>&1
python $root/tools/archive/time_concurrent_ipa.py > $out/${R}_concurrent_ipa.txt 2>&1
python $root/tools/archive/time_poseidon_latency.py > $out/${R}_poseidon_latency_raw.txt 2>&1
for m in fresh contexts-alive threads-leftover torch-first; do python $root/tools/archive/diag_queues2.py $m 2>&1 | grep -v amdgpu.ids; done > $out/${R}_concurrency_after.txt
(python $root/tools/step_breakdown.py 26; python $root/tools/step_breakdown.py 21; echo "# REEF_SC_DEFER=0 (the first fold written out at once, round 3 form):"; REEF_SC_DEFER=0 python $root/tools/step_breakdown.py 26) > $out/${R}_step_breakdown.txt 2>&1
(echo "# tools/time_fold.py: four waves per group operation (shipped up to 2^14 outputs)"; python $root/tools/archive/time_fold.py 8 12 14 15; echo "# REEF_MSM_FOLD_COOP=0: one wave"; REEF_MSM_FOLD_COOP=0 python $root/tools/archive/time_fold.py 8 12 14 15) > $out/${R}_fold_timing.txt 2>&1
(echo "# python tools/step_breakdown.py, the small rounds of a step: shipped"; python $root/tools/step_breakdown.py 21; python $root/tools/step_breakdown.py 26
 echo "# REEF_SC_SPLIT_MAX=0: every dense round with a thread per item (an item's four folds, three products and three reductions one after the other)"; REEF_SC_SPLIT_MAX=0 python $root/tools/step_breakdown.py 21
 echo "# REEF_SC_SPLIT_MAX=0 REEF_SC_ITEMS=1 REEF_SC_FLOOR=1: and a pair per thread in the mid-sized rounds (the round 3 grid)"; REEF_SC_SPLIT_MAX=0 REEF_SC_ITEMS=1 REEF_SC_FLOOR=1 python $root/tools/step_breakdown.py 21) > $out/${R}_small_rounds_ab.txt 2>&1
(cd /tmp; for L in 21 26; do rm -rf /tmp/tl$L; rocprofv3 --kernel-trace --output-format csv -d /tmp/tl$L -- python $root/tools/step_breakdown.py $L > /dev/null 2>&1; echo "## ell = $L: kernels of the last step (us)"; python $root/tools/archive/step_timeline.py /tmp/tl$L/*/*kernel_trace.csv; done) > $out/${R}_step_timeline.txt 2>&1
(export CHUNKS=0,5,8,10,16; python $root/tools/archive/sweep_chunk.py 15 13; python $root/tools/archive/sweep_chunk.py 16 15) > $out/${R}_chunk_sweep.txt 2>&1
# round 5 additions
fi   # MODE = all
sha=$(cd $root && python -c "from reef_amd import _ffi; print(_ffi.library_sources_sha16())")
if [ "$MODE" = all ]; then
$root/reef_amd/_lib/seam_bench > $out/${R}_stateless_concurrent.txt 2>&1
python $root/tools/time_stateless.py --threads 1,4,8 > $out/${R}_stateless_concurrent_python.txt 2>&1
python $root/tools/time_group.py > $out/${R}_group_timing.txt 2>&1
(for m in 2 8; do python $root/bench.py --gpus $m --single-process --steps 3 --warmup 1 --msms-per-step 24 2>/dev/null; done) > $out/${R}_bench_single_process.jsonl
(for m in 3 8; do $root/reef_amd/_lib/reef_replay cfg4 nofold devices=$m 2>/dev/null; done; $root/reef_amd/_lib/reef_replay cfg4b nofold 2>/dev/null) > $out/${R}_replay_devices.jsonl
$root/reef_amd/_lib/affine_probe > $out/${R}_affine_probe.txt 2>&1
bash $root/tools/archive/sweep_sc_mid.sh $out/${R}_sc_mid_sweep.txt
python $root/tools/archive/pmc_sc_step.py 26 $out/${R}_pmc_sc_step_26.json > /dev/null 2>&1; python $root/tools/archive/pmc_sc_step.py 21 $out/${R}_pmc_sc_step_21.json > /dev/null 2>&1
fi   # MODE = all
if [ "$MODE" = final ]; then
python $root/tools/pmc_traffic.py $out > $out/${R}_pmc_traffic.log 2>&1
mkdir -p $root/profiles; cp $out/${R}_pmc_traffic.json $root/profiles/ 2>/dev/null      # bench.py reads roofline.traffic from profiles/ (refused when the kernel sources differ)
python $root/bench.py --steps 20 --warmup 5 > $out/${R}_bench.json 2> $out/${R}_bench.err
(cd /tmp && rm -rf /tmp/ks && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-replay > $out/${R}_bench_under_rocprof.json 2>/dev/null; cp /tmp/ks/*/*kernel_stats.csv $out/${R}_kernel_stats.csv)
(for m in 3 8; do $root/reef_amd/_lib/reef_replay cfg4 nofold devices=$m 2>/dev/null; done; $root/reef_amd/_lib/reef_replay cfg4b nofold 2>/dev/null;
 $root/reef_amd/_lib/reef_replay cfg5 nofold devices=8 2>/dev/null) > $out/${R}_replay_devices.jsonl
python $root/tools/time_group.py > $out/${R}_group_timing.txt 2>&1
python $root/tools/time_merkle.py 16 20 24 26 27 > $out/${R}_merkle_timing.txt 2>&1
python $root/tools/archive/sweep_window_mid.py > $out/${R}_window_mid_sweep.txt 2>&1
REEF_ROUND=$R python $root/tools/pmc_replay.py $out cfg3 > /dev/null 2>&1; REEF_ROUND=$R python $root/tools/pmc_replay.py $out cfg4 > /dev/null 2>&1; REEF_ROUND=$R python $root/tools/pmc_replay.py $out cfg5 > /dev/null 2>&1
python $root/bench.py --gpus 2 --single-process --steps 3 --warmup 1 --msms-per-step 12 > $out/${R}_bench_single_process.jsonl 2>/dev/null
python $root/bench.py --gpus 8 --single-process --steps 3 --warmup 1 --msms-per-step 12 >> $out/${R}_bench_single_process.jsonl 2>/dev/null
python $root/bench.py --gpus 8 --single-process --group-exchange rccl --steps 3 --warmup 1 --msms-per-step 12 2>/dev/null | grep "^{" >> $out/${R}_bench_single_process.jsonl
fi
# the one-launch sum-check rounds under load, ten times the GPU suite's count, all three orderings of the hand-over (sumcheck_kernels.inc: SC_ORDER_*)
(echo "# reef_amd/_lib/sc_stress <ell> <steps> load: one folding step repeated under k_accum0 + streaming load, every coefficient triple against the two-launch form; library sources $sha"
 run() { env REEF_SC_FENCE=$1 $3 $root/reef_amd/_lib/sc_stress $2 $4 load; }
 for f in 0 1 2; do n=$([ $f = 0 ] && echo 20000 || echo 2000)
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
python $root/tools/soak.py ${SOAK_SECONDS:-600} 5 > $out/${R}_soak.txt 2>&1
