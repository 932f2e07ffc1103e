#!/usr/bin/env python3
"""Hyrax-shaped document commitment (HyraxPC::commit, commitment.rs:187) at BASELINE sizes:
   python tools/time_rows.py rows row_len bound [reps]   e.g. cfg3: 1024 2048 131, cfg4: 4096 8192 7"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from reef_amd import msm
from oracle.pasta_oracle import CURVES
rows, row_len, bound = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
k0, d = 1234567, 89
bases = msm.gen_bases("pallas", k0, d, row_len, device=True)
sc = msm.gen_scalars("pallas", 0xD0C, rows * row_len, kind=2, small_bound=bound, mont=True, device=True)
out = msm.DeviceBuffer(96 * rows)
ctx = msm.MsmContext("pallas", bases, row_len)
ctx.enable_timing(True)
ctx.msm(sc, min(row_len, 1024)); ctx.sync()                      # first-launch costs out of the way
t0 = time.perf_counter()
ctx.msm_rows(sc, rows, row_len, max_scalar_bits=bound.bit_length(), out=out); ctx.sync()
first = time.perf_counter() - t0                                  # includes building the block tables, if that path is taken
t0 = time.perf_counter()
for _ in range(reps):
    ctx.msm_rows(sc, rows, row_len, max_scalar_bits=bound.bit_length(), out=out)
ctx.sync()
dt = (time.perf_counter() - t0) / reps
res = out.to_host((rows, 12))
# size-independent check on a few rows: sum_j Z[r,j]*(k0 + j*d) * G
C = CURVES["pallas"]
canon = msm.gen_scalars("pallas", 0xD0C, rows * row_len, kind=2, small_bound=bound, mont=False)[:, 0].astype(object).reshape(rows, row_len)
comp = msm.compress("pallas", res)
ok = True
w = (k0 + np.arange(row_len, dtype=object) * d)
for r in (0, 1, rows // 2, rows - 1):
    acc = int((canon[r] * w).sum()) % C.order
    ok &= comp[32 * r:32 * r + 32] == C.compress(C.mul(acc, C.gen))
sym = msm.DeviceBuffer.from_host(np.ascontiguousarray(msm.gen_scalars("pallas", 0xD0C, rows * row_len, kind=2, small_bound=bound, mont=False)[:, 0].astype(np.uint8)))
ctx.msm_rows_symbols(sym, rows, row_len, max(1, (bound - 1).bit_length()), out=out); ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.msm_rows_symbols(sym, rows, row_len, max(1, (bound - 1).bit_length()), out=out)
ctx.sync()
dts = (time.perf_counter() - t0) / reps
print(f"rows={rows} row_len={row_len} bound={bound}: first call {first*1e3:.3f} ms, then {dt*1e3:.3f} ms per commit from field elements, "
      f"{dts*1e3:.3f} ms from one-byte symbols, {rows*row_len/dt/1e6:.1f} M symbols/s, check={'ok' if ok else 'MISMATCH'} {ctx.timing_stats()}")

# the blinded commitment with everything on the device (blinds and h too): the table of h is checked on the device
# (k_h_refresh), so the call returns without waiting for the GPU
bl = msm.gen_scalars("pallas", 0xB1, rows, device=True)
hh = msm.gen_bases("pallas", 777, 1, 1, device=True)
ctx.msm_rows_symbols(sym, rows, row_len, max(1, (bound - 1).bit_length()), blinds=bl, h=hh, out=out); ctx.sync()
t0 = time.perf_counter()
for _ in range(reps):
    ctx.msm_rows_symbols(sym, rows, row_len, max(1, (bound - 1).bit_length()), blinds=bl, h=hh, out=out)
t_issue = (time.perf_counter() - t0) / reps
ctx.sync()
dtb = (time.perf_counter() - t0) / reps
print(f"  blinded, blinds and h device-resident: {dtb*1e3:.3f} ms per commit from one-byte symbols; the call returns after {t_issue*1e3:.3f} ms")
