#!/usr/bin/env python3
"""Time the sum-check vector kernels (row N2) at Reef's table sizes and relate them to the HBM
roofline:  python tools/time_sumcheck.py [ell ...]   (cfg3: 21, cfg4 hybrid: 26)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from reef_amd import msm
from reef_amd.sumcheck import SumCheck
from oracle.sumcheck_oracle import Q

for ell in [int(x) for x in sys.argv[1:]] or [21, 24, 26]:
    n = 1 << ell
    doc = msm.gen_scalars("pallas", 0xD0C, n, kind=0, mont=False, device=True)
    eqv = msm.gen_scalars("pallas", 0xE9, n, kind=0, mont=False, device=True)
    with SumCheck("pallas", ell) as sc:
        for rep in range(2):                      # second pass is the timed one (first warms allocations)
            sc.set_table_device(0, doc.ptr, n)
            sc.set_table_device(1, eqv.ptr, n)
            sc.sync()
            t_coeff = t_fold = 0.0
            first = None
            t_all = time.perf_counter()
            for i in range(1, ell + 1):
                t0 = time.perf_counter()
                xsq, x, con = sc.round_coeffs(i)
                t1 = time.perf_counter()
                sc.fold(i, (xsq * 7 + 3) % Q)     # stand-in for the Poseidon challenge
                sc.sync()
                t2 = time.perf_counter()
                t_coeff += t1 - t0
                t_fold += t2 - t1
                if i == 1:
                    first = (t1 - t0, t2 - t1)
            total = time.perf_counter() - t_all
        # fused form: one pass per round
        sc.set_table_device(0, doc.ptr, n); sc.set_table_device(1, eqv.ptr, n); sc.sync()
        t_f = time.perf_counter()
        xsq, x, con = sc.round_coeffs(1)
        t_first = None
        for i in range(1, ell + 1):
            rch = (xsq * 7 + 3) % Q
            t0 = time.perf_counter()
            if i < ell:
                xsq, x, con = sc.fold_and_next_coeffs(i, rch)
            else:
                sc.fold(i, rch); sc.sync()
            if i == 1:
                t_first = time.perf_counter() - t0
        total_fused = time.perf_counter() - t_f
        print(f"ell={ell}: fused fold+coeffs: all rounds {total_fused*1e3:.2f} ms; round 1 {t_first*1e3:.3f} ms = {96*n/t_first/1e9:.0f} GB/s", flush=True)
        # the table shaped like Reef's hybrid table (r1cs.rs:481-484, :2105-2112): transitions and one repeated value in the first
        # half, document symbols and zeros in the second; REEF_SC_STRUCT=0 reads it as dense field elements all the same
        import numpy as np
        cols = 1 << (ell // 2)
        hyb = np.zeros((n, 4), dtype=np.uint64)
        rngn = np.random.default_rng(7)
        hyb[:3 * cols + 17] = rngn.integers(0, 1 << 62, size=(3 * cols + 17, 4), dtype=np.uint64)
        hyb[3 * cols + 17:n // 2] = np.array([0x123456789abcdef1, 0x0fedcba987654321, 0x1111111122222222, 0x0333333344444444], dtype=np.uint64)
        hyb[n // 2:n - n // 16, 0] = rngn.integers(0, 7, size=n // 2 - n // 16, dtype=np.uint64)
        d_hyb = msm.DeviceBuffer.from_host(hyb)
        del hyb
        for mode in ("1", "0"):
            os.environ["REEF_SC_STRUCT"] = mode
            sc.set_table_device(0, d_hyb.ptr, n)
            best = None
            for rep in range(3):
                sc.reset_table(); sc.sync()
                t0 = time.perf_counter()
                sc.gen_eq_table([(0x1234567 * (k + 3)) % Q for k in range(34)], [(0x9E3779B1 * (k + 1)) % n for k in range(33)], [(0x7654321 * (k + 5)) % Q for k in range(ell)])
                t_eq = time.perf_counter() - t0
                xsq, x, con = sc.round_coeffs(1)
                t_r1 = time.perf_counter() - t0 - t_eq
                t_f1 = None
                for i in range(1, ell + 1):
                    rch = (xsq * 7 + 3) % Q
                    t1 = time.perf_counter()
                    if i < ell:
                        xsq, x, con = sc.fold_and_next_coeffs(i, rch)
                    else:
                        sc.fold(i, rch); sc.sync()
                    if i == 1:
                        t_f1 = time.perf_counter() - t1
                tot = time.perf_counter() - t0
                if best is None or tot < best[0]:
                    best = (tot, t_eq, t_r1, t_f1, sc.read(0, 1)[0])
            print(f"ell={ell}: one folding step on a hybrid-shaped table, {'row structure used' if mode == '1' else 'read as dense field elements'}: {best[0]*1e3:.2f} ms "
                  f"(round-1 sums {best[2]*1e3:.3f}, first fused round {best[3]*1e3:.3f}); T~(r) = {best[4] & 0xffffffff:08x}", flush=True)
        os.environ.pop("REEF_SC_STRUCT", None)
        d_hyb.free()
        sc.set_table_device(0, doc.ptr, n)
        # one folding step as wit_nlookup_gadget runs it (r1cs.rs:2320-2376): T from its pristine copy, gen_eq_table, ell rounds.
        # REEF_SC_RANK1=0 is the dense form of rounds 1-2 (EQ written out and streamed beside T)
        nq = 33
        rs = [(0x1234567 * (k + 3)) % Q for k in range(nq + 1)]
        qs = [(0x9E3779B1 * (k + 1)) % n for k in range(nq)]
        lq = [(0x7654321 * (k + 5)) % Q for k in range(ell)]
        for mode in ("1", "0"):
            os.environ["REEF_SC_RANK1"] = mode
            best = None
            for rep in range(3):
                sc.reset_table(); sc.sync()
                t0 = time.perf_counter()
                sc.gen_eq_table(rs, qs, lq)
                t_eq = time.perf_counter() - t0
                xsq, x, con = sc.round_coeffs(1)
                t_r1 = time.perf_counter() - t0 - t_eq
                t_f1 = None
                for i in range(1, ell + 1):
                    rch = (xsq * 7 + 3) % Q
                    t1 = time.perf_counter()
                    if i < ell:
                        xsq, x, con = sc.fold_and_next_coeffs(i, rch)
                    else:
                        sc.fold(i, rch); sc.sync()
                    if i == 1:
                        t_f1 = time.perf_counter() - t1
                tot = time.perf_counter() - t0
                if best is None or tot < best[0]:
                    best = (tot, t_eq, t_r1, t_f1, sc.read(0, 1)[0])
            print(f"ell={ell}: one folding step, EQ {'rank-one (never written out)' if mode == '1' else 'dense'}: {best[0]*1e3:.2f} ms "
                  f"(gen_eq_table {best[1]*1e3:.3f}, round-1 sums {best[2]*1e3:.3f}, first fused round {best[3]*1e3:.3f}); T~(r) = {best[4] & 0xffffffff:08x}", flush=True)
        os.environ.pop("REEF_SC_RANK1", None)
        gb_c, gb_f = 64 * n / 1e9, 96 * n / 1e9    # round 1: coeffs read 2 tables, fold reads 2 and writes half
        print(f"ell={ell}: all {ell} rounds {total*1e3:.2f} ms (coeffs {t_coeff*1e3:.2f}, folds {t_fold*1e3:.2f}); "
              f"round 1: coeffs {first[0]*1e3:.3f} ms = {gb_c/first[0]:.0f} GB/s, fold {first[1]*1e3:.3f} ms = {gb_f/first[1]:.0f} GB/s "
              f"(HBM peak 8000 GB/s)", flush=True)

# CPU beside it: the oracle's C port of one round (single thread, Montgomery tables), 2^21 entries
import numpy as np
from oracle import pasta_ref as R
ell = 21
n = 1 << ell
T = msm.gen_scalars("pallas", 0xD0C, n, kind=0, mont=False)
E = msm.gen_scalars("pallas", 0xE9, n, kind=0, mont=False)
R.sc_to_mont(1, T); R.sc_to_mont(1, E)
t0 = time.perf_counter()
R.sc_round(1, T, E, n // 2, 12345)
dt = time.perf_counter() - t0
print(f"CPU port (oracle/pasta_ref.c, 1 thread) round 1 at ell={ell}: {dt*1e3:.1f} ms = {160*n/dt/1e9:.2f} GB/s of table traffic")
