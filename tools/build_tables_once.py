import sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reef_amd import msm
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 15
n = 1 << logn
g = msm.gen_bases("pallas", 3, 5, n, device=True)
msm.MsmContext("pallas", g, 2048, bucket_groups=1, byte_tables=1).close()     # warm-up: module load, first allocations
for tables in (2, 1):
    t = time.perf_counter()
    ctx = msm.MsmContext("pallas", g, n, bucket_groups=1, byte_tables=tables)
    print(f"2^{logn} points, byte_tables={tables}: reef_msm_ctx_create {1e3 * (time.perf_counter() - t):.2f} ms")
    ctx.close()
