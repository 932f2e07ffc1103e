#!/bin/bash
# Round 6 probe: does leaving some SIMDs with ONE k_accum0 wave (184 VGPRs: a 224-VGPR tail kernel fits beside it; beside two it does not) help the tails
# of the other MSMs in flight?  bench.py's timed region, chunk length (entries per k_accum0 thread: 128 = two waves on every SIMD, the shipped plan) x streams.
cd ${GRAFT_REPO_ROOT:-.}
run() { python bench.py --steps 4 --warmup 2 --msms-per-step 48 --no-cpu-baseline --no-replay "$@" 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; s=r['single_stream']
print('%-28s %.4f ms/MSM  k_accum0 in flight %.3f  alone %.3f (msm %.3f)  %s' % ('$*', d['config']['ms_per_msm'], r['kernel_ms'], s['kernel_ms'], s['msm_ms'], d['config']['check']))"; }
for p in $(seq ${1:-3}); do
for c in 0 132 136 144 152 160 176 192 208; do run --chunk $c --streams 3; done
for c in 144 160 192; do run --chunk $c --streams 4; done
done
