#!/usr/bin/env python3
"""Round 4: which state of a long-lived process slows the three concurrent arguments of the replay (bench.py read 9-10 ms in the
driver's run where the harness alone reads 6)?  Each mode is one process:  python tools/diag_queues2.py <mode>"""
import os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from reef_amd import msm, replay

def rep(tag):
    g = replay.run("cfg3", nofold=True, tables=False)
    print(f"{mode:28s} {tag:52s} three at once {g['three_arguments_concurrently_ms']:.2f} ms (one by one {g['ipa_pallas_ms'] + g['ipa_vesta_ms'] + g['consistency_ipa_ms']:.2f})", flush=True)

mode = sys.argv[1] if len(sys.argv) > 1 else "fresh"
n = 1 << 18
if mode == "fresh":
    rep("nothing before")
elif mode == "scratch-under-load":
    # the main thread's scratch stream (reef_gen_bases ...) is created while three contexts are alive, then they are closed
    bases = msm.gen_bases("pallas", 5, 3, n, device=True)
    ctx0 = msm.MsmContext("pallas", bases, n, bucket_groups=1)
    cl = [ctx0.clone() for _ in range(2)]
    sc = msm.gen_scalars("pallas", 1, n, device=True)
    for c in [ctx0] + cl:
        c.msm(sc, n); c.sync()
    for c in [ctx0] + cl:
        c.close()
    rep("3 contexts used and closed")
    rep("again")
elif mode == "contexts-alive":
    bases = msm.gen_bases("pallas", 5, 3, n, device=True)
    ctx0 = msm.MsmContext("pallas", bases, n, bucket_groups=1)
    cl = [ctx0.clone() for _ in range(2)]
    sc = msm.gen_scalars("pallas", 1, n, device=True)
    for c in [ctx0] + cl:
        c.msm(sc, n); c.sync()
    rep("3 idle contexts alive")
    for c in [ctx0] + cl:
        c.close()
    rep("closed")
elif mode == "torch-first":
    import torch
    torch.cuda.set_device(0)
    x = torch.zeros(1 << 20, device="cuda"); torch.cuda.synchronize()
    rep("torch initialised, one tensor")
elif mode == "threads-leftover":
    # six caller threads, each with a clone (the host-scalar leg of the bench), joined; their contexts closed
    bases = msm.gen_bases("pallas", 5, 3, n, device=True)
    ctx0 = msm.MsmContext("pallas", bases, n, bucket_groups=1)
    cl = [ctx0.clone() for _ in range(6)]
    hs = msm.gen_scalars("pallas", 1, n)
    def work(c):
        for _ in range(3):
            c.msm(hs)
    ts = [threading.Thread(target=work, args=(c,)) for c in cl]
    [t.start() for t in ts]; [t.join() for t in ts]
    for c in cl + [ctx0]:
        c.close()
    rep("6 caller threads joined, contexts closed")
elif mode == "sumcheck-first":
    from reef_amd.sumcheck import SumCheck
    with SumCheck("pallas", 20) as s:
        s.set_table(0, [1, 2, 3]); s.sync()
    rep("a sum-check context created and destroyed")
elif mode == "many-streams":
    # 40 contexts created and destroyed: where does the runtime's queue assignment stand afterwards?
    bases = msm.gen_bases("pallas", 5, 3, 4096, device=True)
    for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 5):
        c = msm.MsmContext("pallas", bases, 4096, bucket_groups=1); c.close()
    rep("contexts created and destroyed one by one")
