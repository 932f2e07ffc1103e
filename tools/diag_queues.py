import os, sys, threading
sys.path.insert(0, ".")
import numpy as np, torch
from reef_amd import msm, replay
def rep(tag):
    g = replay.run("cfg3", nofold=True, tables=False); print(tag, g["three_arguments_concurrently_ms"], flush=True)
torch.cuda.set_device(0)
rep("fresh")
n = 1 << 18
bases = msm.gen_bases("pallas", 5, 3, n, device=True)
sc = msm.gen_scalars("pallas", 3, n, device=True)
ctx0 = msm.MsmContext("pallas", bases, n, bucket_groups=1)
ctxs = [ctx0] + [ctx0.clone() for _ in range(2)]
outs = [msm.DeviceBuffer(96) for _ in ctxs]
for _ in range(20):
    for c, o in zip(ctxs, outs): c.msm(sc, n, out=o)
for c in ctxs: c.sync()
rep("after 3 contexts (alive)")
more = [ctx0.clone() for _ in range(6)]
hs = msm.gen_scalars("pallas", 3, n)
def w(c):
    for _ in range(5): c.msm(hs, n)
th = [threading.Thread(target=w, args=(c,)) for c in more]
[t.start() for t in th]; [t.join() for t in th]
rep("after 6 more clones (alive)")
for c in more: c.close()
rep("6 clones closed")
for c in ctxs: c.close()
rep("all closed")
from reef_amd.sumcheck import SumCheck
with SumCheck("pallas", 20) as s: s.set_table(0, [1,2,3])
rep("after a sum-check context")
x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
rep("after torch work")
import bench
bench.hbm_bound_leg(24)
rep("after bench.hbm_bound_leg(24)")
rep("again")
pin = torch.empty((n, 4), dtype=torch.int64).pin_memory()
rep("after pinning 8 MB")
