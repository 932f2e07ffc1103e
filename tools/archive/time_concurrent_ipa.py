#!/usr/bin/env python3
"""How do latency-bound IPA rounds from several caller threads share the GPU?  N threads, each with its own resident key of
2^logn generators, each running `rounds` cross-term rounds back to back:   python tools/time_concurrent_ipa.py [logn] [rounds]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from reef_amd import msm

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 15
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = 1 << logn
R = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001
gens = msm.gen_bases("pallas", 5, 3, n, device=True)
a = msm.gen_scalars("pallas", 3, n)
ctxs = [msm.MsmContext("pallas", gens, n, bucket_groups=1, byte_tables=2) for _ in range(4)]
for c in ctxs:
    c.ipa_cross_terms(a, [], [])

def work(c):
    for _ in range(rounds):
        c.ipa_cross_terms(a, [], [])

base = None
for nt in (1, 2, 3, 4):
    best = 1e9
    for rep in range(3):
        th = [threading.Thread(target=work, args=(ctxs[i],)) for i in range(nt)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        best = min(best, time.perf_counter() - t0)
    base = base or best
    print(f"2^{logn} generators, {nt} caller thread(s) x {rounds} rounds: {best*1e3:.2f} ms = {best/rounds*1e3:.3f} ms per round per thread, "
          f"{nt*rounds/best:.0f} rounds/s ({nt*base/best:.2f}x one thread)", flush=True)
