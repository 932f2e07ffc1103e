#!/usr/bin/env python3
"""What makes the three concurrent arguments of the replay slow inside a long-lived process?  (bench.py reads 9-10 ms where the
harness alone reads 6.)  python tools/diag_queues.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from reef_amd import msm, replay
def rep(tag):
    g = replay.run("cfg3", nofold=True, tables=False); print(tag, g["three_arguments_concurrently_ms"], g["setup_ms"], flush=True)
torch.cuda.set_device(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "churn-first"
if mode == "replay-first":
    rep("fresh")
big = [msm.DeviceBuffer(2 << 30) for _ in range(3)]
for b in big: b.free()
rep("after allocating and freeing 3 x 2 GiB")
n = 1 << 20
bases = msm.gen_bases("pallas", 5, 3, n, device=True)
ctx0 = msm.MsmContext("pallas", bases, n, bucket_groups=1)      # 16 tables of 64 MiB
rep("with a 2^20-point pre-shifted key alive")
ctx0.close()
rep("key closed")
rep("again")
