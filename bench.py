#!/usr/bin/env python3
"""bench.py -- MSM scalar-point pairs/s of the MI355X backend (BASELINE.json metric).

A step = one pass of the hot path over one batch of synthetic input: a batch of 256 MSMs of 2^20 Pallas points each
(BASELINE.json configs[1]; --msms-per-step) with the key and the scalars already resident in HBM, issued one by one over
three streams.  (Rounds 1-2 counted ONE MSM as a step, round 3 a batch of 32: the driver's 20 steps were 26 ms, then 0.8 s of a
13 s run, and its sampler never saw the GPU busy; with 256 they are ~6 s.  `config.ms_per_msm` is the figure rounds 1-2 called
ms_per_step; `value` -- pairs per second -- does not depend on the batch.)
    python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches one rank per GPU with torch.distributed.run; every rank owns its
own 2^20 points of one N*2^20-point MSM (weak scaling), the 96-byte partial sums are exchanged
with an RCCL all-gather and added on the device (RCCL has no elliptic-curve reduction).
Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events on the stream the
accumulation kernel is launched on; `cpu_baseline` times the oracle's C restatement of the CPU
Pippenger on ALL host cores (rank 0, N = 1 only, bounded sample) -- it is the checker, never the
product path.

What `value` includes: key AND scalars are resident in HBM when the timed region starts
(`config.scalars` says so); the PCIe-inclusive rate of the same workload -- scalars in pinned host
memory, result back in host memory, the same MSMs in flight -- is measured after the timed region and
reported as `config.host_scalars_ms_per_msm`; it is never `value`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
BYTES_PER_PAIR = 96            # 32 B scalar + 64 B affine base, each read once
FMUL_PEAK = 1.57e11            # Montgomery products/s chip-wide, measured (reef_bench_fmul, profiles/README.md)


import threading as _threading
WATCHDOG_FIRED = _threading.Event()     # set by the N > 1 watchdog of the side legs (main)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20, help="timed steps; a step is one batch of --msms-per-step MSMs")
    p.add_argument("--msms-per-step", type=int, default=256, help="MSMs of 2^logn points per step, issued one by one over the streams (the batch of synthetic "
                                                                  "input one step passes through the hot path).  256 x 1.2 ms = 0.3 s a step, so that the driver's "
                                                                  "20 steps are a timed region of ~6 s: long enough for an external sampler at 5 s intervals to see "
                                                                  "the GPU busy (rounds 1-3: 0 of 3 samples)")
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--logn", type=int, default=20, help="log2 of points per GPU (BASELINE configs[1]: 20)")
    p.add_argument("--curve", default="pallas")
    p.add_argument("--scalars", default="uniform", choices=["uniform", "witness"])
    p.add_argument("--window-bits", type=int, default=0)
    p.add_argument("--bucket-groups", type=int, default=-1, help="-1: engine default for the bench key")
    p.add_argument("--chunk", type=int, default=0)
    p.add_argument("--streams", type=int, default=3, help="MSMs in flight (clones of the key on separate HIP streams)")
    p.add_argument("--batch", type=int, default=1, help="N = 1 only: MSMs per step, issued as ONE batched call on the same resident key (reef_msm_rows with "
                                                        "rows = batch: one pass of the sort / accumulate / reduce pipeline for all of them)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    p.add_argument("--no-check", action="store_true")
    p.add_argument("--allow-experiment", action="store_true",
                   help="report `value` although the library loaded is the +experiment build (REEF_MSM_LIB=.../libreef_msm_exp.so: A/B runs); without it the "
                        "bench refuses: the measured artefact is the release build (reef_amd/_lib/libreef_msm.so)")
    p.add_argument("--no-replay", action="store_true", help="skip the replay of Reef's own MSM sequence (config.replay_cfg3) after the timed region")
    p.add_argument("--exercise-collective", action="store_true",
                   help="with --gpus 1: run the N > 1 code path (process group of one rank, all_gather + on-device combine) "
                        "to check the RCCL/stream plumbing on a single-GPU box")
    p.add_argument("--sharding", default="points", choices=["points", "windows"],
                   help="N > 1: 'points' = every GPU owns its own 2^logn pairs of one N*2^logn-point MSM (weak scaling, default); "
                        "'windows' = ONE 2^logn-point MSM, GPU r accumulates the Pippenger windows w = r mod N (strong scaling)")
    p.add_argument("--single-process", action="store_true",
                   help="with --gpus N and NO torchrun: ONE process drives N members through the library's device groups (reef_msm_group_*, "
                        "include/reef_msm.h section 5) -- what a Rust prover can call; members beyond the visible devices repeat ordinals (labelled)")
    p.add_argument("--group-exchange", default="peer", choices=["peer", "host", "rccl"],
                   help="--single-process: how the members' 96-byte partial sums reach devices[0] (include/reef_msm.h section 5: hipMemcpyPeerAsync, host-mapped slots, or "
                        "ncclSend/ncclRecv on a single-process RCCL communicator)")
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="nccl = RCCL over xGMI (default); gloo = host-staged gather (debug / boxes without RCCL), labelled as such")
    return p.parse_args()


def dlog_of_msm(curve_name, canon, k0, d, offset):
    """sum_i s_i * (k0 + (offset+i)*d) mod the group order: the discrete log of an MSM over bases in
    arithmetic progression, with O(n) big-int work."""
    import numpy as np
    from oracle.pasta_oracle import CURVES
    C = CURVES[curve_name]
    n = canon.shape[0]
    idx = np.arange(offset, offset + n, dtype=object)
    acc = 0
    for j in range(4):
        col = canon[:, j].astype(object)
        acc += (int(col.sum()) * k0 + int((col * idx).sum()) * d) << (64 * j)
    return acc % C.order


def point_of_dlog(curve_name, k):
    from oracle.pasta_oracle import CURVES
    C = CURVES[curve_name]
    return C.compress(C.mul(k % C.order, C.gen))


def cpu_quota():
    """CPUs the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited: os.cpu_count() shows the host's."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def cpu_baseline(curve_id, seconds):
    """Oracle C Pippenger on the host cores: the window-parallel form on a persistent thread pool (pasta-msm's shape: one
    window size for the whole input, Booth digits, (window, point-slice) tiles dealt out to the pool)."""
    from oracle import pasta_ref as R
    cores = os.cpu_count() or 1
    n = 1 << 18
    bases = R.gen_bases_ap(curve_id, 3, 5, n)
    sc = R.gen_scalars(curve_id, 0x5EEF, n)
    R.msm_pippenger_windows(curve_id, bases, sc, threads=cores)          # creates the pool's threads
    # os.cpu_count() is what the box shows, not what the container is given (a CPU quota, SMT siblings): the thread count is
    # chosen by a probe, and the speed-up over one thread is reported next to it
    t0 = time.perf_counter()
    R.msm_pippenger_windows(curve_id, bases, sc, threads=1, n=n // 8)
    one_thread_rate = (n // 8) / (time.perf_counter() - t0)
    quota = cpu_quota()
    if quota:      # threads beyond the quota only burn the period's budget sooner and are throttled for the rest of it
        cand = {min(cores, max(1, int(quota))), min(cores, max(1, int(2 * quota))), min(cores, max(1, int(quota) // 2))}
    else:
        cand = {cores, max(1, cores // 2), max(1, cores // 4)}
    best = None
    for t in sorted(cand):                                               # ties go to the smaller count
        t0 = time.perf_counter()
        for _ in range(3):                                               # long enough to average over the quota's periods
            R.msm_pippenger_windows(curve_id, bases, sc, threads=t)
        dt = (time.perf_counter() - t0) / 3
        if best is None or dt < 0.97 * best[1]:
            best = (t, dt)
    threads, dt = best
    rate = n / dt
    logn = 18
    while logn < 20 and (1 << (logn + 1)) / rate * 3 < seconds:          # the bench's own size when the budget allows
        logn += 1
    if logn != 18:
        n = 1 << logn
        bases = R.gen_bases_ap(curve_id, 3, 5, n)
        sc = R.gen_scalars(curve_id, 0x5EEF, n)
    reps, spent = 0, 0.0
    while spent < seconds * 0.5 or reps < 2:
        t0 = time.perf_counter()
        R.msm_pippenger_windows(curve_id, bases, sc, threads=threads)
        spent += time.perf_counter() - t0
        reps += 1
    c, slices = R.window_plan(n, threads)
    value = n * reps / spent
    return {"value": value, "unit": "pairs/s", "cores": threads, "kind": "port", "per_thread": value / threads,
            "one_thread": one_thread_rate, "speedup_over_one_thread": value / one_thread_rate, "host_cores_visible": cores,
            "container_cpu_quota": quota,
            "pool_helpers": R.pool_size(),
            "sample": f"{reps} x 2^{logn}-point Pallas MSM, uniform scalars, oracle/pasta_ref.c window-parallel Pippenger (c = {c}, {slices} point "
                      f"slices per window, persistent thread pool; a restatement, NOT the reference binary: Reef is Rust and cannot be built here), "
                      f"{threads} threads on {cores} visible host cores (the container's CPU quota is {quota if quota else 'none'}: best of quota/2, quota, 2 x quota threads -- or of all, 1/2, 1/4 of the cores without one; speed-up over one thread {value / one_thread_rate:.1f}x)"}


def hbm_bound_leg(ell=26):
    """After the timed region, never `value`: the HBM-bound row of the same path under the driver's clock -- one fused round of the
    nlookup sum-check (row N2: fold with the previous challenge and sum the next round's coefficients in ONE pass,
    src/backend/r1cs_helper.rs:441-506) over a table of 2^ell field elements, as a folding step of wit_nlookup_gadget runs it
    (r1cs.rs:2320-2376: gen_eq_table, then the rounds).  Since round 3 the EQ table is never written out while the rounds fold its
    high index bits (it stays two factor tables plus point masses), so the pass streams T only; the dense form (both tables
    streamed, rounds 1-2) is timed beside it on a caller-given EQ.  Check: the sum-check identity of the round."""
    from reef_amd import msm
    from reef_amd.sumcheck import SumCheck
    Q = msm.PALLAS_SCALAR_Q
    n = 1 << ell
    doc = msm.gen_scalars("pallas", 0xD0C, n, kind=0, mont=False, device=True)   # full-width entries: no row of the table is constant or small,
    eqv = msm.gen_scalars("pallas", 0xE9, n, kind=0, mont=False, device=True)   # so round one streams 32-byte entries like every later round
    nq = 33
    rs = [(0x1234567 * (k + 3)) % Q for k in range(nq + 1)]
    qs = [(0x9E3779B1 * (k + 1)) % n for k in range(nq)]
    lq = [(0x7654321 * (k + 5)) % Q for k in range(ell)]
    with SumCheck("pallas", ell) as sc:
        best = dense = None
        ok = True
        sc.set_table_device(0, doc.ptr, n)
        for rep in range(3):
            for form in ("rank-one", "dense"):
                sc.reset_table()
                if form == "dense":
                    sc.set_table_device(1, eqv.ptr, n)       # a caller-given EQ is a dense table
                else:
                    sc.gen_eq_table(rs, qs, lq)
                sc.sync()
                xsq, x, con = sc.round_coeffs(1)
                r = (xsq * 7 + 3) % Q
                claim = (xsq * r * r + x * r + con) % Q          # g_1(r): what round 2 must sum to
                t0 = time.perf_counter()
                xsq2, x2, con2 = sc.fold_and_next_coeffs(1, r)   # returns after the pass (the coefficients come back to the host)
                dt = time.perf_counter() - t0
                ok = ok and (con2 + (xsq2 + x2 + con2)) % Q == claim   # g_2(0) + g_2(1)
                if form == "dense":
                    dense = dt if dense is None or dt < dense else dense
                else:
                    best = dt if best is None or dt < best else best
    doc.free()
    eqv.free()
    algo = n * 32 + (n // 2) * 32                            # T read once, half of it written
    algo_dense = 2 * n * 32 + 2 * (n // 2) * 32              # both tables read once, half of each written
    return {"kernel": "k_sc_r1_fold_coeffs (fused sum-check round, row N2; EQ kept as factor tables)", "table_entries": n, "entry_bytes": 32,
            "round_ms": best * 1e3, "algorithmic_bytes": algo, "achieved": algo / best / 1e9, "unit": "GB/s", "peak": HBM_PEAK_GBS,
            "frac": algo / best / 1e9 / HBM_PEAK_GBS, "bound": "hbm", "check": "sumcheck-identity-ok" if ok else "MISMATCH",
            "dense_form": {"kernel": "k_sc_fold_coeffs (both tables streamed)", "round_ms": dense * 1e3, "algorithmic_bytes": algo_dense,
                           "achieved": algo_dense / dense / 1e9, "frac": algo_dense / dense / 1e9 / HBM_PEAK_GBS},
            "note": "host-timed call (launches, pass, 96-byte read-back included); PMC traffic of these kernels equals their algorithmic bytes "
                    "(profiles/r03_pmc_streaming.json)"}


def _diag(tag):
    """REEF_BENCH_DIAG=1: how the final SNARK's three arguments share the GPU at this point of the process (stderr)."""
    at = os.environ.get("REEF_BENCH_DIAG_AT")          # one probe at the named point only (the first replay of the process)
    if not (os.environ.get("REEF_BENCH_DIAG") == "1" or (at and at == tag)):
        return
    from reef_amd import replay
    g = replay.run("cfg3", nofold=True, tables=False)
    print(f"[diag] {tag}: three arguments at once {g['three_arguments_concurrently_ms']} ms", file=sys.stderr, flush=True)


SAFA_CAVEAT = ("MSM lengths are PREDICTIONS of Reef's cost model (src/backend/costs.rs restated: oracle/costs_oracle.py), not measurements of a Reef run; since round 5 the "
               "automaton behind every config is DERIVED (oracle/safa_shape.py restates SAFA::new for skips and literals; tests/test_safa_shape.py) from a regex with a "
               "source -- cfg3: '.*password.*' (BASELINE configs[2]): 12 states, 1292 edges; cfg4: tests/scripts/dna.sh:8 re-based to a 16 MiB document: 64 states, 309 edges, "
               "2 folding steps at -b 32 (tests/golden/replay_shapes.json `inputs`, `hand_count`)")


def replay_leg(cfg="cfg3", with_tables=True):
    """After the timed region, never `value`: the MSM sequence of one `reef --prove` on BASELINE.json configs[2] (cfg3, the 1 MiB
    document) or configs[3] (cfg4: north_star's own target, the 16 MiB DNA document with --hybrid -b 32)
    (src/backend/framework.rs:664-723) issued in-process through the C ABI by the C++ harness (per-step scalars in host
    memory, commitments back to the host, every one checked).  Runs BEFORE the CPU baseline: the sequence is bound by
    host latency, and a container that has just burnt its CPU quota on the oracle's threads is throttled."""
    from reef_amd import replay
    out = {}
    g = replay.run(cfg, nofold=True, tables=False)
    out.update({"workload": g["replay"], "shapes": "tests/golden/replay_shapes.json (Reef's cost model restated: oracle/costs_oracle.py)",
                "shapes_caveat": SAFA_CAVEAT,
                "w1": g["w1"], "c1": g["c1"], "w2": g["w2"], "c2": g["c2"], "steps": g["steps"],
                "fold_ms_per_step": g["ms_per_step"], "fold_ms_per_step_batched_pairs": g["ms_per_step_batched_pairs"],
                "fold_ms_per_step_concurrent": g.get("ms_per_step_concurrent"),
                "ipa_ms": g["ipa_pallas_ms"] + g["ipa_vesta_ms"], "consistency_ipa_ms": g["consistency_ipa_ms"],
                "three_arguments_concurrently_ms": g.get("three_arguments_concurrently_ms"),
                "three_arguments_note": "the two Spartan arguments and the consistency argument from three caller threads at once (one after the other: ipa_ms + "
                                        "consistency_ipa_ms); since round 4 the library's contexts share one pool of streams by activity, so the figure no longer "
                                        "depends on what else the process has alive (profiles/r04_concurrency_bisect.txt: 9.6 -> 6.0 ms in this process)",
                "sumcheck_ms_per_step": g.get("sumcheck_ms_per_step"), "sumcheck_table_log": g.get("sumcheck_table_log"),
                "total_prove_msm_ms": g["total_prove_msm_ms"], "total_prove_gpu_ms": g["total_prove_gpu_ms"], "setup_ms": g["setup_ms"],
                "setup_first_ms": g.get("setup_first_ms"), "setup_path": g.get("setup_path"),
                # Reef re-derives its keys on every --prove (framework.rs:115,297-303): set-up IS prove time, so the figure to quote is this one
                "prove_gpu_incl_setup_ms": g["setup_ms"] + g["total_prove_gpu_ms"],
                "commitments_checked_against_dlog": g["commitments_checked_against_dlog"],
                "scalars": "host memory in, commitments back to the host (PCIe-inclusive)", "ipa": g["ipa"]})
    if with_tables:
        try:
            t = replay.run(cfg, nofold=True, tables=True)
            out["byte_tables"] = {"fold_ms_per_step": t["ms_per_step"], "ipa_ms": t["ipa_pallas_ms"] + t["ipa_vesta_ms"],
                                  "three_arguments_concurrently_ms": t.get("three_arguments_concurrently_ms"),
                                  "total_prove_msm_ms": t["total_prove_msm_ms"], "setup_ms": t["setup_ms"],
                                  "prove_gpu_incl_setup_ms": t["setup_ms"] + t["total_prove_gpu_ms"],
                                  "commitments_checked_against_dlog": t["commitments_checked_against_dlog"]}
            # the byte tables cost set-up and save time per folding step: from how many steps on does a run that builds them come out ahead?
            saved_per_step = g["ms_per_step"] - t["ms_per_step"]
            fixed = (t["setup_ms"] - g["setup_ms"]) - ((g["total_prove_gpu_ms"] - g["fold_steps_ms"]) - (t["total_prove_gpu_ms"] - t["fold_steps_ms"]))
            out["byte_tables"]["break_even_steps"] = (max(0.0, fixed) / saved_per_step) if saved_per_step > 0 else None
            out["byte_tables"]["break_even_note"] = ("folding steps after which building the byte tables has paid for itself: (extra set-up - what the final SNARK saves) / "
                                                     "(ms saved per step); this config folds %d times" % g["steps"])
        except Exception as e:
            out["byte_tables"] = {"error": str(e)}
    return out


def replay_cpu_leg(out, cfg="cfg3", cpu_threads=None):
    """The same sequence on the host cores through the oracle (test infrastructure used as the reported CPU side)."""
    from reef_amd import replay
    from oracle import replay_cpu
    c = replay_cpu.run(cfg, replay.SHAPES_PATH, cpu_threads or os.cpu_count() or 1)
    out.update({"cpu_restatement_ms": c["total_prove_msm_ms"], "cpu_fold_ms_per_step": c["ms_per_step"],
                "cpu_ipa_ms": c["ipa_pallas_ms"] + c["ipa_vesta_ms"], "cpu_consistency_ipa_ms": c["consistency_ipa_ms"], "cores": c["threads"], "cpu_kind": c["kind"]})


def independent_units_legs(a, rank, world, dist):
    """After the timed region of an N > 1 run, never `value`.  (1) final_snark_ms: three inner-product arguments without generator folds
    (reef_ipa_cross_terms on resident keys of cfg4's sizes: 2^16 Pallas, 2^14 Vesta, 2^13 Hyrax row generators) dealt out whole by
    reef_amd.distributed.place_units, one all-gather of their L/R points; beside it rank 0 running all three one after the other.
    (2) sumcheck_ms_per_step: one folding step over a 2^26-entry table (cfg4's ell) sharded by low index bits
    (LowBitShardedSumCheck): rank-local folds, 96 B of partial coefficients per rank and round."""
    import numpy as np
    import torch
    from reef_amd import msm
    from reef_amd import distributed as D
    from reef_amd.sumcheck import SumCheck
    import datetime
    # a rank that fails before a collective must not leave the others waiting for the default half hour
    hg = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=180))
    out = {}

    def max_over_ranks(sec):
        t = torch.tensor([sec], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=hg)
        return float(t.item())

    # ---- (1) the three arguments
    sizes = [("pallas", 1 << 16), ("vesta", 1 << 14), ("pallas", 1 << 13)]
    costs = [7.2, 4.4, 3.7]                                   # ms alone on one MI355X (profiles/r03_replay_prove_msm.jsonl, cfg4)
    owner = D.place_units(costs, world)
    rounds = [n.bit_length() - 1 for _, n in sizes]
    widths = [r * 24 for r in rounds]
    keys, vecs = {}, {}
    for u, (cv, n) in enumerate(sizes):
        if owner[u] == rank or rank == 0:                     # rank 0 also holds all three for the one-GPU figure
            bases = msm.gen_bases(cv, 0xC0FFEE + u, 7, n, device=True)
            keys[u] = msm.MsmContext(cv, bases, n, bucket_groups=1)
            vecs[u] = msm.gen_scalars(cv, 40 + u, n, kind=0, mont=True)

    def run_unit(u):
        ctx, v, res = keys[u], vecs[u], []
        w1s, w2s, ln = [], [], sizes[u][1]
        while ln > 1:
            L, R = ctx.ipa_cross_terms(v[:ln], w1s, w2s)
            res += [L, R]
            w1s.append(0x1234567890abcdef + len(w1s)); w2s.append(0x0badc0ffee0ddf00 + len(w2s))
            ln //= 2
        return np.concatenate(res)
    for u in keys:
        run_unit(u)                                           # warm-up: workspaces
    dist.barrier(group=hg)
    t0 = time.perf_counter()
    got = D.run_placed_units(run_unit, widths, costs, group=hg)
    placed = max_over_ranks(time.perf_counter() - t0)
    one_gpu = None
    if rank == 0:
        t0 = time.perf_counter()
        ref = [run_unit(u) for u in range(3)]
        one_gpu = time.perf_counter() - t0
        # the points computed on other ranks are the ones rank 0 computes itself (compared in the canonical encoding: Jacobian
        # representatives depend on the order the sort's atomics happened to take)
        same = all(msm.compress(sizes[u][0], ref[u].reshape(-1, 12)) == msm.compress(sizes[u][0], got[u].reshape(-1, 12)) for u in range(3))
    else:
        same = True
    out["final_snark"] = {"arguments": [f"{cv} 2^{n.bit_length() - 1}" for cv, n in sizes], "owner": owner, "ms": placed * 1e3,
                          "one_gpu_one_after_the_other_ms": one_gpu * 1e3 if one_gpu else None, "check": "same-points" if same else "MISMATCH",
                          "note": "three whole arguments dealt out by cost (place_units), one host-staged all-gather of their L/R points; driven from Python with the "
                                  "vectors in host memory (the C++ harness reads 7.2 + 4.4 + 3.7 ms one after the other on one GPU)"}
    for k in keys.values():
        k.close()

    # ---- (2) one sum-check step, table sharded by low index bits
    ell = 26
    kbits = world.bit_length() - 1
    if 1 << kbits == world:
        Q = msm.PALLAS_SCALAR_Q
        loc = ell - kbits
        with SumCheck("pallas", loc) as eng:
            shard = msm.gen_scalars("pallas", 0xD0C + rank, 1 << loc, kind=0, mont=False, device=True)   # the rank's entries i = rank (mod N)
            eng.set_table_device(0, shard.ptr, 1 << loc)
            sh = D.LowBitShardedSumCheck(eng, ell, Q, group=hg)
            nq = 33
            rs = [(0x1234567 * (k + 3)) % Q for k in range(nq + 1)]
            qs = [(0x9E3779B1 * (k + 1)) % (1 << ell) for k in range(nq)]
            last_q = [(0x7654321 * (k + 5)) % Q for k in range(ell)]
            chal = lambda i, xsq, x, con: (con * 3 + x * 5 + xsq * 7 + i) % Q      # stands in for the Poseidon sponge (same on every rank)
            best, ident = None, True
            for rep in range(3):
                eng.reset_table()
                dist.barrier(group=hg)
                t0 = time.perf_counter()
                sh.gen_eq_table(rs, qs, last_q)
                coeffs, cs, t_fin, e_fin = sh.run_step(chal)
                dt = max_over_ranks(time.perf_counter() - t0)
                best = dt if best is None else min(best, dt)
                claim = None                                   # the sum-check identity round after round: g_i(0) + g_i(1) = g_{i-1}(r_{i-1})
                for (xsq, x, con), r in zip(coeffs, cs):
                    if claim is not None:
                        ident = ident and (2 * con + x + xsq) % Q == claim
                    claim = (xsq * r * r + x * r + con) % Q
                ident = ident and claim == t_fin * e_fin % Q
            shard.free()
        out["sumcheck"] = {"ell": ell, "entries_per_rank": 1 << loc, "ms_per_step": best * 1e3, "check": "sumcheck-identity-ok" if ident else "MISMATCH",
                           "note": "gen_eq_table + all rounds; folds are rank-local, a round exchanges 3 x 32 B per rank (host-staged: the challenge is the host's)"}
    else:
        out["sumcheck"] = {"skipped": "the low-bit shard needs a power-of-two number of ranks"}
    return out


def single_process_main(a):
    """`bench.py --gpus N --single-process`: the multi-GPU split as ONE process sees it -- no torch.distributed, no launcher: the C ABI's
    device groups (reef_msm_group_*).  Reef's prover is one Rust process (src/backend/main.rs:82, framework.rs:81-166); this is the line
    its integration would produce.  Timed region (`value`): weak scaling by points, as the torchrun path -- one N*2^logn-point MSM per
    call over a points group (every member owns 2^logn pairs), three groups in flight from three caller threads, the scalars resident
    on devices[0] (members on other devices fetch their slice over xGMI inside the call: labelled).  After it, the same strong_scaling
    fields as the torchrun line: ONE 2^logn-point MSM split by window (north_star) and by points, and configs[3]'s Hyrax rows dealt
    out whole; every combined point is checked against its discrete logarithm."""
    import threading
    import numpy as np
    from reef_amd import msm
    if os.environ.get("REEF_MSM_HW_QUEUES", "") != "0":
        os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("REEF_MSM_HW_QUEUES") or "8")
    visible = msm.device_count()
    if visible < 1:
        raise SystemExit("bench.py --single-process: no HIP device visible (there is no CPU fallback)")
    N = a.gpus
    devices = [i % visible for i in range(N)]
    n = 1 << a.logn
    k0, d = 0xABCDEF, 0x12345
    kind = 0 if a.scalars == "uniform" else 1
    groups_opt = a.bucket_groups if a.bucket_groups >= 0 else 1
    T = max(1, a.streams)
    MPS = max(1, a.msms_per_step)
    msm.set_device(devices[0])
    EX = {"peer": msm.EXCHANGE_PEER, "host": msm.EXCHANGE_HOST, "rccl": msm.EXCHANGE_RCCL}[a.group_exchange]

    def in_flight(calls, fns):
        """`calls` calls dealt round-robin to len(fns) caller threads; -> seconds"""
        per = [calls // len(fns) + (1 if j < calls % len(fns) else 0) for j in range(len(fns))]
        errs = []

        def work(j):
            try:
                for _ in range(per[j]):
                    fns[j]()
            except BaseException as e:
                errs.append(repr(e))
        th = [threading.Thread(target=work, args=(j,)) for j in range(len(fns))]
        t0 = time.perf_counter()
        [x.start() for x in th]
        [x.join() for x in th]
        dt = time.perf_counter() - t0
        if errs:
            raise RuntimeError(errs[0])
        return dt

    # ---- the timed region: weak scaling by points
    bases_all = msm.gen_bases(a.curve, k0, d, N * n, device=True)
    sc_all = msm.gen_scalars(a.curve, 0x5EEF, N * n, kind=kind, mont=True, device=True)
    wg = [msm.MsmGroup(a.curve, bases_all, devices, N * n, split=msm.SPLIT_POINTS, exchange=EX, window_bits=a.window_bits, bucket_groups=groups_opt, chunk=a.chunk)
          for _ in range(T)]
    info = wg[0].info()
    last = [None] * T

    def weak_call(j):
        def f():
            last[j] = wg[j].msm(sc_all, N * n)
        return f
    wf = [weak_call(j) for j in range(T)]
    in_flight(max(T, a.warmup * MPS), wf)
    elapsed = in_flight(a.steps * MPS, wf)
    weak_ms = elapsed / (a.steps * MPS) * 1e3
    # the same calls with the scalars in (pageable) HOST memory, as a Rust Vec is: every member uploads its own slice over its own link
    hs_all = sc_all.to_host((N * n, 4))
    hf = [(lambda j: (lambda: wg[j].msm(hs_all)))(j) for j in range(T)]
    in_flight(T, hf)
    hcalls = max(T, min(a.steps * MPS, 24))
    host_ms = in_flight(hcalls, hf) / hcalls * 1e3
    check_ok = True
    if not a.no_check:
        canon = msm.gen_scalars(a.curve, 0x5EEF, N * n, kind=kind, mont=False)
        want_all = point_of_dlog(a.curve, dlog_of_msm(a.curve, canon, k0, d, 0))
        check_ok = all(msm.compress(a.curve, r) == want_all for r in last if r is not None) and msm.compress(a.curve, wg[0].msm(hs_all)) == want_all
    for g in wg:
        g.close()

    # ---- one GPU, the same protocol: what one member needs for a 2^logn-point MSM with T calls in flight
    bases1 = msm.gen_bases(a.curve, k0, d, n, device=True)
    sc1 = msm.gen_scalars(a.curve, 0x5EEF, n, kind=kind, mont=True, device=True)
    c0 = msm.MsmContext(a.curve, bases1, n, window_bits=a.window_bits, bucket_groups=groups_opt, chunk=a.chunk, device=devices[0])
    cs = [c0] + [c0.clone() for _ in range(T - 1)]
    plan = c0.plan()
    for c_ in cs:
        c_.enable_timing(True)
    outs = [np.zeros(12, dtype=np.uint64) for _ in cs]
    of = [(lambda j: (lambda: cs[j].msm(sc1, n, out=outs[j])))(j) for j in range(T)]
    in_flight(T, of)
    for c_ in cs:
        c_.timing_stats(reset=True)
    ocalls = max(T, min(a.steps * MPS, 48))
    one_gpu_ms = in_flight(ocalls, of) / ocalls * 1e3
    st = [c_.timing_stats(reset=True) for c_ in cs]
    acc_ms = sum(x["accumulate_ms"] for x in st) / max(1, sum(x["calls"] for x in st))
    want1 = None
    if not a.no_check:
        canon1 = msm.gen_scalars(a.curve, 0x5EEF, n, kind=kind, mont=False)
        want1 = point_of_dlog(a.curve, dlog_of_msm(a.curve, canon1, k0, d, 0))
        check_ok = check_ok and msm.compress(a.curve, outs[0]) == want1
    for c_ in cs:
        c_.close()

    # ---- strong scaling: ONE 2^logn-point MSM over the members, by window and by points; latency (one call in flight) and T in flight
    strong = {"one_msm_points": n, "in_flight": T, "one_gpu_ms_per_msm": one_gpu_ms}
    for name, sp in (("windows", msm.SPLIT_WINDOWS), ("points", msm.SPLIT_POINTS)):
        gs = [msm.MsmGroup(a.curve, bases1, devices, n, split=sp, exchange=EX, window_bits=a.window_bits, bucket_groups=groups_opt, chunk=a.chunk) for _ in range(T)]
        res = [None] * T
        gf = [(lambda j: (lambda: res.__setitem__(j, gs[j].msm(sc1, n))))(j) for j in range(T)]
        in_flight(T, gf)
        calls = max(T, min(a.steps * MPS, 24))
        strong[name + "_ms_per_step"] = in_flight(calls, gf) / calls * 1e3
        strong[name + "_latency_ms"] = in_flight(max(3, calls // T), gf[:1]) / max(3, calls // T) * 1e3
        if want1 is not None:
            check_ok = check_ok and all(msm.compress(a.curve, r) == want1 for r in res)
        for g in gs:
            g.close()
    strong["speedup_vs_1"] = {"windows": one_gpu_ms / strong["windows_ms_per_step"], "points": one_gpu_ms / strong["points_ms_per_step"]}
    # ---- where one split call's time goes (round 6: the first run on several physical devices cannot be rehearsed, so the line itemises it and sets the
    # model's figure beside each term): one call in flight, medians of 9 calls, reef_msm_group_last_timing
    import statistics
    hs1 = sc1.to_host((n, 4))

    def phases(group, scal):
        group.enable_timing(True)
        rows = []
        for _ in range(11):
            r = group.msm(scal, n)
            rows.append(group.last_timing())
        group.enable_timing(False)
        rows = rows[2:]
        med = lambda f: statistics.median(f(t) for t in rows)                                       # noqa: E731
        return {"total_ms": med(lambda t: t["total_ms"]), "scalar_distribution_ms": med(lambda t: t["distribute_ms"]),
                "member_msm_ms": {"max": med(lambda t: max(t["member_stream_ms"])), "min": med(lambda t: min(t["member_stream_ms"]))},
                "member_issue_ms": {"max": med(lambda t: max(t["member_issue_ms"])), "min": med(lambda t: min(t["member_issue_ms"]))},
                "members_finish_after_distribution_ms": med(lambda t: max(0.0, t["members_done_ms"] - t["distribute_ms"])),
                "exchange_ms": med(lambda t: t["combine_ms"]),
                "exchange_note": "the members' 96-byte sends ride on their own streams (inside member_msm_ms); exchange_ms is what is left once every member has finished: "
                                 "the sum of the partial sums on devices[0] and its way to the host"}, r
    # DESIGN.md 7 "Predicted scaling": stage costs of one 2^20-point MSM alone on one GPU (us): recode 32, sort 260 of which 60 do not shrink with the digits,
    # accumulation 1050, merge 70, bucket reduction 160, exchange + N-1 additions 25; scaled to this run's one-GPU latency
    alone_ms = strong.get("windows_latency_ms") if N == 1 else None
    model_one = 32 + 260 + 1050 + 70 + 160
    pred = {"windows_latency_ms": (32 + 60 + (200 + 1050) / N + 70 + 160 + 25) / 1e3 * (n / (1 << 20) if n < (1 << 20) else 1.0),
            "points_latency_ms": (0.30 + (model_one / 1e3 - 0.30) / N + 0.025),
            "basis": "DESIGN.md 7 (stage costs of one 2^20-point MSM alone on one GPU from profiles/r02_msm_kernel_timelines.txt; the host scalars' upload is NOT in the "
                     "model: 32 MiB over one PCIe link ~0.6 ms pinned, 2-4 ms pageable -- compare scalar_distribution_ms)"}
    ph = {"predicted": pred, "protocol": "one call in flight, medians of 9 timed calls (reef_msm_group_enable_timing: every member is waited for separately before the sum)"}
    for name, sp in (("windows", msm.SPLIT_WINDOWS), ("points", msm.SPLIT_POINTS)):
        for mode_name, mode in (("each_member_uploads", msm.SCALARS_EACH), ("fanout_from_member_0", msm.SCALARS_FANOUT)):
            if sp == msm.SPLIT_POINTS and mode == msm.SCALARS_FANOUT:
                continue
            try:
                with msm.MsmGroup(a.curve, bases1, devices, n, split=sp, exchange=EX, window_bits=a.window_bits, bucket_groups=groups_opt, chunk=a.chunk, scalars=mode) as g:
                    g.msm(hs1)
                    g.msm(sc1, n)
                    entry = {}
                    entry["host_scalars"], r_h = phases(g, hs1)
                    if mode == msm.SCALARS_EACH:
                        entry["device_scalars"], r_d = phases(g, sc1)
                        if want1 is not None:
                            check_ok = check_ok and msm.compress(a.curve, r_d) == want1
                    if want1 is not None:
                        check_ok = check_ok and msm.compress(a.curve, r_h) == want1
                    ph[name + ("" if mode == msm.SCALARS_EACH else "_" + mode_name)] = entry
            except Exception as e:                              # instrumentation never takes the line down
                ph[name + "_" + mode_name] = {"error": str(e)}
    strong["phases"] = ph
    strong["note"] = ("ONE 2^logn-point MSM over the members of a device group of THIS process: windows = member i accumulates the Pippenger windows "
                      "w = i (mod N) on a replicated key (north_star's split), points = contiguous slices; *_ms_per_step with `in_flight` groups called from as "
                      "many threads, *_latency_ms with one; one_gpu_ms_per_msm is one context per call on devices[0] under the same protocol")
    # independent units: configs[3]'s Hyrax commitment, rows dealt out whole (document in host memory, as Reef holds it)
    h_rows, h_len, h_bits, hk0, hd = 4096, 8192, 3, 0xFEED, 3
    if a.logn >= 16:
        doc = np.random.default_rng(0xD0C).integers(0, 7, size=(h_rows, h_len), dtype=np.uint8)
        hb = msm.gen_bases(a.curve, hk0, hd, h_len, device=True)
        with msm.MsmGroup(a.curve, hb, devices, h_len, split=msm.SPLIT_WINDOWS, bucket_groups=0) as hg:
            flat = np.ascontiguousarray(doc.reshape(-1))
            got = hg.msm_rows_symbols(flat, h_rows, h_len, h_bits)          # builds the symbol tables
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                got = hg.msm_rows_symbols(flat, h_rows, h_len, h_bits)
            h_ms = (time.perf_counter() - t0) / reps * 1e3
        from oracle.pasta_oracle import CURVES
        order = CURVES[a.curve].order
        rows_ok = True
        for r_ in (0, h_rows // max(N, 1) - 1, h_rows // 2, h_rows - 1):
            dl = sum(int(v) * (hk0 + j * hd) for j, v in enumerate(doc[r_].tolist())) % order
            rows_ok = rows_ok and msm.compress(a.curve, got[r_].copy()) == point_of_dlog(a.curve, dl)
        check_ok = check_ok and rows_ok
        strong["hyrax_rows"] = {"rows": h_rows, "row_len": h_len, "symbol_bits": h_bits, "ms_per_commit": h_ms, "check": "dlog-ok" if rows_ok else "MISMATCH",
                                "note": "HyraxPC::commit of a 16 MiB DNA document from host bytes, rows dealt out in contiguous blocks to the members, results written "
                                        "straight into the caller's array (PCIe-inclusive)"}
    # independent units: configs[4]'s --merkle commitment, the Poseidon tree in blocks over the devices (stand-in constants: timing and equality
    # with one device's root; hash parity is tests/test_gpu_merkle.py)
    if a.logn >= 16:
        from reef_amd import merkle
        # the permutation's constants are the caller's (neptune's, on the Rust side); here: seeded round constants and a Cauchy matrix 1 / (i + 5 + j) over
        # Pallas' scalar field, width 5, 8 full and 56 partial rounds (the shape of neptune's U4 instance)
        FQ = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
        prng = np.random.default_rng(0x9051D0)
        m_rc = [int.from_bytes(prng.bytes(40), "little") % FQ for _ in range(5 * (8 + 56))]
        m_mds = [[pow(i + 5 + j, -1, FQ) for j in range(5)] for i in range(5)]
        mlog = 24
        mdoc = np.random.default_rng(0x3E2C).integers(0, 256, size=1 << mlog, dtype=np.uint32)
        margs = ("pallas", mdoc, 5, 8, 56, m_rc, m_mds, 4 << 32 | 1, 2 << 32 | 1)
        root1, _ = merkle.commit_arrays(*margs, want_tree=False)
        t0 = time.perf_counter()
        root1, _ = merkle.commit_arrays(*margs, want_tree=False)
        one_ms = (time.perf_counter() - t0) * 1e3
        minfo = {}
        rootd, _ = merkle.commit_arrays(*margs, want_tree=False, devices=devices, info=minfo)
        t0 = time.perf_counter()
        rootd, _ = merkle.commit_arrays(*margs, want_tree=False, devices=devices, info=minfo)
        dev_ms = (time.perf_counter() - t0) * 1e3
        check_ok = check_ok and rootd == root1
        strong["merkle_commit"] = {"symbols": 1 << mlog, "blocks": minfo.get("blocks"), "ms_per_commit": dev_ms, "one_device_ms_per_commit": one_ms,
                                   "check": "same-root" if rootd == root1 else "MISMATCH",
                                   "note": "MerkleCommitment::new (merkle_tree.rs:25-80) of a 2^24-symbol document from host memory, the bottom level cut into power-of-two "
                                           "blocks, one per device, the levels above hashed from the blocks' roots on devices[0]; root returned (PCIe-inclusive)"}
    value = N * n * a.steps * MPS / elapsed
    achieved = BYTES_PER_PAIR * n / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else None
    out = {"metric": "msm_scalar_point_pairs_per_sec", "value": value, "unit": "pairs/s", "n_gpus": N, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
           "config": {"workload": f"one {N} x 2^{a.logn}-point {a.curve.capitalize()} MSM per call over a device group of {N} members (2^{a.logn} pairs per member), "
                                  f"{a.scalars} 255-bit scalars; a step = {MPS} such calls, {T} groups in flight",
                      "mode": "single-process: reef_msm_group_* of libreef_msm.so (no torch, no launcher) -- what a Rust prover calls",
                      "devices": devices, "distinct_devices": info["distinct_devices"], "visible_devices": visible,
                      "devices_note": None if info["distinct_devices"] == N else f"{N} members on {info['distinct_devices']} device(s): ordinals repeat, NOT a scaling measurement",
                      "exchange": info["exchange"] + (" (hipMemcpyPeerAsync of the 96-byte partial sums to devices[0], sum kernel there; in place where members share a device)"
                                                     if info["exchange"] == "peer" else
                                                     " (ncclSend/ncclRecv pairs in one ncclGroup on a single-process communicator, RCCL opened at run time; sum kernel on devices[0])"
                                                     if info["exchange"] == "rccl" else ""),
                      "peer_members": info["peer_members"], "key_points_per_member": info["key_points"],
                      "scalars": "device-resident on devices[0]; members on other devices fetch their slice with a peer copy INSIDE the timed call",
                      "ms_per_msm": weak_ms, "host_scalars_ms_per_msm": host_ms,
                      "host_scalars_note": "the same calls with the scalars in pageable host memory (a Rust Vec): every member uploads its slice over its own PCIe link",
                      "points_per_gpu": n, "total_points": N * n, "window_bits": plan["window_bits"], "windows": plan["windows"], "bucket_groups": plan["bucket_groups"],
                      "tables": plan["tables"], "streams": T, "msms_per_step": MPS, "sharding": "points", "strong_scaling": strong,
                      "check": ("dlog-ok" if check_ok else "MISMATCH") if not a.no_check else "skipped"},
           "roofline": {"bound": "hbm", "kernel": "k_accum0 (bucket accumulation)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": None, "kernel_ms": acc_ms,
                        "algorithmic_bytes_per_launch": BYTES_PER_PAIR * n,
                        "note": f"measured on the one-GPU leg of this run (HIP events on the contexts' streams, {T} MSMs in flight); the group's members run the same kernel"},
           "cpu_baseline": None}
    print(json.dumps(out), flush=True)
    if not a.no_check and not check_ok:
        sys.exit(2)


def main():
    a = parse()
    # RCCL between processes (and CUDA-tensor sharing) needs dmabuf IPC on this pool's host driver; the boxes export it already -- kept here so that a
    # launcher with a scrubbed environment does not end in `hipIpcGetMemHandle: invalid argument`.  Before anything initialises HIP; children inherit it.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if a.single_process:
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise SystemExit("bench.py --single-process runs WITHOUT torch.distributed.run: one process drives all the devices")
        return single_process_main(a)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU (torch.distributed.run --nproc-per-node {a.gpus})")
    # more hardware queues than the runtime's default four, before anything initialises HIP: caller threads' streams that share a
    # queue run one after the other (reef_amd/csrc/api.cpp; the library asks for the same when it is loaded first)
    if os.environ.get("REEF_MSM_HW_QUEUES", "") != "0":
        os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("REEF_MSM_HW_QUEUES") or "8")
    import numpy as np
    import torch
    import torch.distributed as dist
    from reef_amd import msm
    from reef_amd import _ffi as _ffi_mod

    ndev = max(1, msm.device_count())
    dev_index = local_rank % ndev          # one rank per GPU; (gloo debug runs may share a device)
    torch.cuda.set_device(dev_index)
    msm.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    multi = a.gpus > 1 or a.exercise_collective      # the path with an exchange step
    # stdout carries exactly one JSON line: RCCL prints a version banner to the C-level stdout at
    # exit, so fd 1 is pointed at stderr for the life of the process and the line goes to a saved copy
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if multi:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        if dist.get_world_size() != a.gpus:      # the first multi-GPU run must not quietly measure something else
            raise RuntimeError(f"--gpus {a.gpus} but the process group has {dist.get_world_size()} ranks")
    n = 1 << a.logn
    k0, d = 0xABCDEF, 0x12345
    kind = 0 if a.scalars == "uniform" else 1
    by_windows = a.sharding == "windows" and multi
    owner = 0 if by_windows else rank            # window split: every rank holds the same points and scalars
    seed = 0x5EEF + owner
    groups = a.bucket_groups if a.bucket_groups >= 0 else 1   # bench key: full precompute (fixed commitment key)

    # synthetic inputs, generated on the device: rank r owns bases B_i, i in [r*n, (r+1)*n)
    bases = msm.gen_bases(a.curve, k0 + owner * n * d, d, n, device=True)
    B = a.batch if not multi else 1
    scalars = msm.gen_scalars(a.curve, seed, n * B, kind=kind, mont=True, device=True)
    library = _ffi_mod.load().reef_version().decode()
    if "+experiment" in library and not a.allow_experiment:
        raise SystemExit(f"bench.py measures the release build; the library loaded is {library!r} ({_ffi_mod.LIB_PATH}): unset REEF_MSM_LIB or pass --allow-experiment")
    torch.cuda.synchronize()
    t_key = time.perf_counter()
    ctx0 = msm.MsmContext(a.curve, bases, n, window_bits=a.window_bits, bucket_groups=groups, chunk=a.chunk)   # returns with the tables built
    key_build_ms = (time.perf_counter() - t_key) * 1e3
    if by_windows:
        ctx0.set_window_split(rank, max(world, 1))
    _diag("before the contexts")
    ctxs = [ctx0] + [ctx0.clone() for _ in range(max(1, a.streams) - 1)]
    _diag("contexts created")
    plan = ctx0.plan()
    for c_ in ctxs:
        c_.enable_timing(True)        # roofline.achieved needs the accumulation kernel's own duration
    nctx = len(ctxs)
    # per-context device buffers: local partial (96 B), gathered partials, combined result
    parts = [torch.zeros(96, dtype=torch.uint8, device=dev) for _ in range(nctx)]
    gathered = [torch.zeros(96 * a.gpus, dtype=torch.uint8, device=dev) for _ in range(nctx)]
    results = [torch.zeros(96 * B, dtype=torch.uint8, device=dev) for _ in range(nctx)]
    # the exchange of N > 1 (reef_amd/distributed.py: all-gather of the 96-byte partial sums + on-device add); with RCCL
    # the collective is ordered on the MSM's own HIP stream, so a step needs no host sync
    from reef_amd.distributed import PartialSumExchange
    exch = None
    if multi:
        exch = []
        for c in ctxs:
            stream_ctx = None
            if a.backend == "nccl":
                try:
                    es = torch.cuda.ExternalStream(c.stream, device=dev)
                    stream_ctx = (lambda es_: (lambda: torch.cuda.stream(es_)))(es)
                except Exception as e:     # older torch: order by a host sync before the collective
                    print(f"[bench] ExternalStream unavailable ({e}); syncing before each all_gather", file=sys.stderr)
            exch.append(PartialSumExchange((lambda c_: (lambda g, cnt, out: c_.sum_points(g.data_ptr(), cnt, out.data_ptr())))(c),
                                           backend=a.backend, before_exchange=c.sync, stream_ctx=stream_ctx))

    def one_msm(i):
        j = i % nctx
        c = ctxs[j]
        if not multi and B > 1:
            c.msm_rows(scalars, B, n, max_scalar_bits=255, out=results[j].data_ptr())
        elif not multi:
            c.msm(scalars, n, out=results[j].data_ptr())
        else:
            c.msm(scalars, n, out=parts[j].data_ptr())
            exch[j].combine(parts[j], gathered[j], results[j])

    def sync_all():
        for c in ctxs:
            c.sync()
        torch.cuda.synchronize()

    if multi and a.backend == "nccl":
        try:                       # first collective on an external stream: fall back to host-ordered calls if torch refuses
            one_msm(0)
            sync_all()
        except Exception as e:
            print(f"[bench] collective on the MSM stream failed ({e}); ordering by host sync instead", file=sys.stderr)
            for x in exch:
                x.stream_ctx = None
    stream_ordered = bool(multi and a.backend == "nccl" and exch and all(x.stream_ctx is not None for x in exch))
    MPS = max(1, a.msms_per_step)

    def step(i):              # one step = one batch of MPS MSMs (each with its exchange when N > 1), round-robin over the streams
        for k in range(MPS):
            one_msm(i * MPS + k)

    for i in range(a.warmup):
        step(i)
    sync_all()
    for c in ctxs:
        c.timing_stats(reset=True)
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    sync_all()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stats = [c.timing_stats(reset=True) for c in ctxs]
    _diag("after the timed region")
    calls = sum(s["calls"] for s in stats)
    acc_ms = sum(s["accumulate_ms"] for s in stats) / max(calls, 1)
    tot_ms = sum(s["total_ms"] for s in stats) / max(calls, 1)
    last = (a.steps * MPS - 1) % nctx
    final_parts = parts[last].cpu().numpy().copy()
    final_result = results[last].cpu().numpy().copy()
    single = None
    host_ms = None
    host_pageable = None
    plain_key = None

    # ---- N > 1: the timed region's own result is checked BEFORE any side leg runs, and the side legs run under a deadline ----
    # Everything from here to the end of the run has never met more than one physical GPU (RCCL on > 1 rank, peer copies between
    # devices, the child process over all devices): if one of its collectives hangs, the measured line must not hang with it.  So
    # (1) the weak region's combined point is compared with its discrete logarithm first, on every rank (weak_check below: the same
    # collectives the check used after the legs until round 5), and (2) a watchdog thread on every rank waits REEF_BENCH_SIDE_TIMEOUT
    # seconds (default 480) for the rest of the run: when they pass, rank 0 prints the line it has -- `value` of the completed timed
    # region, the weak check's verdict, `strong_scaling` = the error -- and every rank leaves through os._exit (a rank stuck in a
    # collective cannot be unwound).  A run that finishes in time never notices any of this.
    import threading
    weak = None                      # (ok, partials_differ, discrete logarithm of rank 0's MSM) once weak_check has run
    line_state = {"lock": threading.Lock(), "done": False, "build": None}
    run_finished = threading.Event()

    def emit(line):
        with line_state["lock"]:
            if line_state["done"]:
                return False
            line_state["done"] = True
        os.write(json_fd, (json.dumps(line) + "\n").encode())
        return True

    def weak_check():
        canon_ = msm.gen_scalars(a.curve, seed, n * B, kind=kind, mont=False)
        my_dlog_ = dlog_of_msm(a.curve, canon_[:n], k0, d, owner * n)
        cdev_ = dev if a.backend == "nccl" else "cpu"
        got_total = msm.compress(a.curve, final_result.view(np.uint64))
        got_part = msm.compress(a.curve, final_parts.view(np.uint64))
        words = torch.tensor([(my_dlog_ >> (32 * j)) & 0xFFFFFFFF for j in range(8)], dtype=torch.int64, device=cdev_)
        allw = [torch.zeros_like(words) for _ in range(a.gpus)]
        dist.all_gather(allw, words)
        dlogs = [sum(int(v) << (32 * j) for j, v in enumerate(w.tolist())) for w in allw]
        if by_windows:                    # every rank holds the same MSM; only the combined point is one
            total_dlog, ok_ = my_dlog_, True
        else:                             # points: the rank's partial is its own slice's MSM, the total is the sum of the slices
            ok_ = got_part == point_of_dlog(a.curve, my_dlog_)
            total_dlog = sum(dlogs)
        ok_ = ok_ and got_total == point_of_dlog(a.curve, total_dlog)     # EVERY rank checks the combined point against the expected total
        differ_ = (got_part != got_total) if a.gpus > 1 else None
        flag = torch.tensor([1 if ok_ else 0, 1 if (differ_ or a.gpus == 1) else 0], device=cdev_)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag[0].item()), (bool(flag[1].item()) if a.gpus > 1 else None), dlogs[0]

    def build_line(check, partials_differ, strong):
        # HBM traffic of the dominant kernel: PMC counters need their own rocprofv3 passes (never
        # combined with timing), so the value is read from the committed profile of this command
        # a profile is taken only if it was collected on the kernels this run executes (sources fingerprint) and on this plan; a stale
        # one is refused rather than quoted (no fall-back to an older round's file)
        traffic, traffic_src = None, None
        from reef_amd import _ffi as _f
        prof_name = "r06_pmc_traffic.json"
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", prof_name)))
            pc = prof["config"]
            same_plan = (pc["curve"], pc["logn"], pc["window_bits"], pc["bucket_groups"]) == (a.curve, a.logn, plan["window_bits"], plan["bucket_groups"])
            if prof.get("kernel_sources_sha16") != _f.kernel_sources_sha16():
                traffic_src = f"REFUSED: profiles/{prof_name} was collected on other kernel sources ({prof.get('kernel_sources_sha16')} != {_f.kernel_sources_sha16()}); re-run tools/pmc_traffic.py"
            elif not same_plan:
                traffic_src = f"REFUSED: profiles/{prof_name} describes another plan ({pc})"
            else:
                accum = [v for k, v in prof["kernels"].items() if k.startswith(f"k_accum0<{msm.curve_id(a.curve)}")]
                traffic = max(v["hbm_bytes_per_launch"] for v in accum)
                traffic_src = f"profiles/{prof_name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on kernel sources {prof['kernel_sources_sha16']})"
        except (OSError, KeyError, ValueError) as e:
            traffic_src = f"no usable profiles/{prof_name}: {e}"
        pairs = n * B * MPS * (1 if by_windows else a.gpus) * a.steps
        value = pairs / elapsed
        achieved = BYTES_PER_PAIR * n * B / (acc_ms * 1e-3) / 1e9 if acc_ms > 0 else None
        eff_windows = min(plan["windows"], -(-255 // plan["window_bits"]))   # windows that hold scalar bits (scalars < 2^255)
        out = {
            "metric": "msm_scalar_point_pairs_per_sec", "value": value, "unit": "pairs/s", "n_gpus": a.gpus,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if by_windows else "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"2^{a.logn}-point {a.curve.capitalize()} MSM per GPU, {a.scalars} 255-bit scalars, "
                                   f"resident key, device-resident scalars (BASELINE.json configs[1]); a step = a batch of {MPS * B} such MSMs; `value` is the "
                                   f"FIXED-KEY figure (commitment keys are fixed for a proof: the key's {plan['tables']} pre-shifted tables are built once, "
                                   f"key_build_ms, outside the timed region); the variable-base figure on the same inputs is config.plain_key",
                       "library": library, "key_build_ms": key_build_ms, "plain_key": plain_key,
                       "scalars": "device-resident (generated on the GPU before the timed region; no PCIe traffic inside it)",
                       "host_scalars_ms_per_msm": host_ms,
                       "host_scalars_pageable_ms_per_msm": host_pageable,
                       "host_scalars_note": "same MSMs with the scalars in host memory and the result returned to the host, caller threads each on its own clone of "
                                            "the resident key; PCIe-inclusive, measured after the timed region, never `value`.  host_scalars_ms_per_msm: PINNED memory, six "
                                            "callers; host_scalars_pageable_ms_per_msm: ordinary pageable memory (what a Rust Vec is), one caller and six",
                       "points_per_gpu": n, "total_points": n * (1 if by_windows else a.gpus), "window_bits": plan["window_bits"],
                       "windows": plan["windows"], "bucket_groups": plan["bucket_groups"], "tables": plan["tables"],
                       "streams": nctx, "msms_per_step": MPS * B, "ms_per_msm": elapsed / (a.steps * MPS * B) * 1e3, "sharding": ("windows (w = rank mod N)" if by_windows else "points") if a.gpus > 1 else "none",
                       "exchange": ("none" if a.gpus == 1 else "rccl all_gather of 96 B partials + on-device add" if a.backend == "nccl"
                                    else "HOST-STAGED gloo all_gather of 96 B partials (debug fallback, not RCCL) + on-device add"),
                       "check": check, "partials_differ_from_total": partials_differ, "msm_ms_stream": tot_ms,
                       "strong_scaling": strong,
                       "rccl": ({"ranks_seen": dist.get_world_size(), "backend": a.backend, "stream_ordered": stream_ordered} if multi else None),
                       "streams_of_the_contexts": (len({c_.stream for c_ in ctxs}) if multi else None)},
            "roofline": {"bound": "hbm", "kernel": "k_accum0 (bucket accumulation)", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "traffic_source": traffic_src,
                         "kernel_ms": acc_ms, "algorithmic_bytes_per_launch": BYTES_PER_PAIR * n * B,
                         "note": f"kernel_ms = average launch duration with {nctx} MSMs in flight (launches stretch each other); single_stream = one MSM in flight",
                         "single_stream": ({"kernel_ms": single["kernel_ms"], "msm_ms": single["msm_ms"],
                                            "achieved": BYTES_PER_PAIR * n / (single["kernel_ms"] * 1e-3) / 1e9,
                                            "frac": BYTES_PER_PAIR * n / (single["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                                           if single and single["kernel_ms"] > 0 else None),
                         # the kernel is bound by integer issue, not HBM: whole-job field products per second
                         # (10 per bucket addition, one addition per non-zero digit ~ windows-1 per pair) against the
                         # measured chip-wide rate of the Montgomery product (reef_bench_fmul)
                         "issue": {"field_products_per_s": value / a.gpus * eff_windows * 10, "peak": FMUL_PEAK,
                                   "frac": value / a.gpus * eff_windows * 10 / FMUL_PEAK, "unit": "products/s"}},
        }
        return out
    line_state["build"] = build_line

    if multi and not a.no_check:
        weak = weak_check()
    if multi and a.gpus > 1:
        side_deadline = float(os.environ.get("REEF_BENCH_SIDE_TIMEOUT", "480"))   # below the 600 s after which torch aborts a rank waiting in an RCCL collective

        def watchdog():
            if run_finished.wait(side_deadline):
                return
            why = (f"the legs after the timed region did not finish within {side_deadline:.0f} s (REEF_BENCH_SIDE_TIMEOUT): one of their collectives or the "
                   "child process hangs; the timed region itself had completed on every rank (closing barrier and max-over-ranks all_reduce returned)")
            WATCHDOG_FIRED.set()       # from here on an exception in the main thread (a peer that has left) waits for this thread instead of ending the process
            print(f"[bench] rank {rank}: {why}", file=sys.stderr)
            if rank == 0 and line_state["build"] is not None:
                verdict = "skipped" if weak is None else ("dlog-ok (timed region; the strong-scaling legs never finished)" if weak[0] else "MISMATCH")
                emit(line_state["build"](verdict, weak[1] if weak else None, {"error": why}))
            sys.stderr.flush()
            time.sleep(1.0)            # the ranks' deadlines lie milliseconds apart: nobody leaves before every rank's watchdog has fired
            os._exit(2 if (weak is not None and not weak[0]) else 0)
        threading.Thread(target=watchdog, daemon=True).start()

    # ---- N > 1, after the timed region (never `value`): the two STRONG splits of ONE 2^logn-point MSM on the same ranks ----
    # (a) by Pippenger window, the split north_star names: every rank holds all points and scalars and accumulates the
    #     windows w = rank (mod N) (reef_msm_ctx_set_window_split); (b) by points: rank r owns pairs [r n/N, (r+1) n/N).
    # Both end in the same all-gather of 96-byte partial sums + on-device add; both results are checked below against
    # the discrete logarithm of the whole MSM.  The weak-scaling region above stays the bench line's `value`.
    strong = None
    strong_results = {}
    if multi and a.gpus > 1 and not by_windows and os.environ.get("REEF_BENCH_STRONG", "1") != "0":
        try:       # a side measurement: a failure that every rank sees alike (an exception raised before any collective) does not cost the bench line
            from reef_amd.distributed import shard_bounds
            cdev0 = dev if a.backend == "nccl" else "cpu"
            ksteps = max(3, min(a.steps * MPS, 24))

            def make_exch(c):
                sc_ = None
                if a.backend == "nccl" and stream_ordered:
                    es_ = torch.cuda.ExternalStream(c.stream, device=dev)
                    sc_ = (lambda e_: (lambda: torch.cuda.stream(e_)))(es_)
                return PartialSumExchange((lambda c_: (lambda g, cnt, out: c_.sum_points(g.data_ptr(), cnt, out.data_ptr())))(c),
                                          backend=a.backend, before_exchange=c.sync, stream_ctx=sc_)

            def timed(cs, xs, sc_ptr, cnt):
                def one(i):
                    j = i % len(cs)
                    cs[j].msm(sc_ptr, cnt, out=parts[j].data_ptr())
                    xs[j].combine(parts[j], gathered[j], results[j])
                def sync_cs():
                    for c in cs:
                        c.sync()
                    torch.cuda.synchronize()
                for i in range(len(cs)):
                    one(i)
                sync_cs()
                dist.barrier()
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                for i in range(ksteps):
                    one(i)
                sync_cs()
                dist.barrier()
                torch.cuda.synchronize()
                t_ = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64, device=cdev0)
                dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                return float(t_.item()) / ksteps * 1e3, results[(ksteps - 1) % len(cs)].cpu().numpy().copy()

            def phases_of(c, x, sc_ptr, cnt):
                """Where ONE split MSM's time goes on this run's ranks (round 6: the first SCALE line must be diagnosable): one call in flight, 9 timed
                repetitions, per rank the host-timed MSM (enqueue to the partial sum finished) and exchange (all-gather of the 96-byte partial sums + the
                on-device sum, to finished); medians per rank, then max / min over the ranks."""
                import statistics
                ms_, ex_ = [], []
                for it in range(11):
                    dist.barrier()
                    t0p = time.perf_counter()
                    c.msm(sc_ptr, cnt, out=parts[0].data_ptr())
                    c.sync()
                    t1p = time.perf_counter()
                    x.combine(parts[0], gathered[0], results[0])
                    c.sync()
                    torch.cuda.synchronize()
                    t2p = time.perf_counter()
                    if it >= 2:
                        ms_.append((t1p - t0p) * 1e3)
                        ex_.append((t2p - t1p) * 1e3)
                mine = torch.tensor([statistics.median(ms_), statistics.median(ex_)], dtype=torch.float64, device=cdev0)
                allp = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(allp, mine)
                m_ = [float(t[0]) for t in allp]
                e_ = [float(t[1]) for t in allp]
                return {"member_msm_ms": {"max": max(m_), "min": min(m_), "per_rank": m_}, "exchange_ms": {"max": max(e_), "min": min(e_)},
                        "note": "exchange_ms contains the wait for the slowest rank's partial sum (a rank that finishes early waits in the all-gather)"}

            # rank 0's points and scalars are THE MSM; the other ranks generate the same ones (seeded, on the device)
            bases0 = bases if rank == 0 else msm.gen_bases(a.curve, k0, d, n, device=True)
            scal0 = scalars if rank == 0 else msm.gen_scalars(a.curve, 0x5EEF, n, kind=kind, mont=True, device=True)
            if rank == 0:
                wfirst = ctx0.clone()
            else:
                wfirst = msm.MsmContext(a.curve, bases0, n, window_bits=a.window_bits, bucket_groups=groups, chunk=a.chunk)
            wfirst.set_window_split(rank, world)
            wctx = [wfirst] + [wfirst.clone() for _ in range(nctx - 1)]          # clones inherit the split
            wx = [make_exch(c) for c in wctx]
            w_ms, w_res = timed(wctx, wx, scal0, n)
            try:
                w_phases = phases_of(wctx[0], wx[0], scal0, n)
            except Exception as e:
                w_phases = {"error": str(e)}
            for c in wctx:
                c.close()
            lo, hi = shard_bounds(n, world, rank)
            bases_s = msm.gen_bases(a.curve, k0 + lo * d, d, hi - lo, device=True)
            pfirst = msm.MsmContext(a.curve, bases_s, hi - lo, window_bits=a.window_bits, bucket_groups=groups, chunk=a.chunk)
            pctx = [pfirst] + [pfirst.clone() for _ in range(nctx - 1)]
            px = [make_exch(c) for c in pctx]
            p_ms, p_res = timed(pctx, px, scal0.ptr + 32 * lo, hi - lo)
            try:
                p_phases = phases_of(pctx[0], px[0], scal0.ptr + 32 * lo, hi - lo)
            except Exception as e:
                p_phases = {"error": str(e)}
            for c in pctx:
                c.close()
            one_gpu_ms = elapsed / (a.steps * MPS) * 1e3      # what one GPU needs for a 2^logn-point MSM in the same regime (the weak region above)
            strong = {"one_msm_points": n, "steps": ksteps, "in_flight": nctx,
                      "windows_ms_per_step": w_ms, "points_ms_per_step": p_ms,
                      "speedup_vs_1": {"windows": one_gpu_ms / w_ms, "points": one_gpu_ms / p_ms},
                      "one_gpu_ms_per_msm": one_gpu_ms,
                      "phases": {"windows": w_phases, "points": p_phases,
                                 "predicted": {"windows_member_msm_ms": (32 + 60 + (200 + 1050) / world + 70 + 160) / 1e3, "points_member_msm_ms": 0.30 + 1.27 / world,
                                               "exchange_ms": 0.025, "windows_ms_per_step_in_flight": {1: 1.33, 2: 0.85, 4: 0.58, 8: 0.45}.get(world),
                                               "basis": "DESIGN.md 7 'Predicted scaling' (stage costs of one 2^20-point MSM on one GPU, us: recode 32, sort 260 of which 60 fixed, "
                                                        "accumulation 1050, merge 70, bucket reduction 160, all-gather + N-1 additions 25)"}},
                      "note": "ONE 2^logn-point MSM split over the ranks (strong scaling), timed after the weak-scaling region with the same barrier + "
                              "max-over-ranks protocol; windows = reef_msm_ctx_set_window_split(rank, N) on replicated points and scalars, points = "
                              "contiguous slices; one_gpu_ms_per_msm is the weak region's time per MSM (a 2^logn-point MSM per GPU); the *_ms_per_step figures here are per MSM"}
            strong_results = {"windows": w_res, "points": p_res}
            # (c) independent units: the Hyrax document commitment of BASELINE.json configs[3] -- 4096 rows of
            #     8192 three-bit symbols over shared row generators (src/backend/commitment.rs:187) -- with the rows dealt out in
            #     contiguous blocks; one all-gather of the 96-byte row commitments.  Sampled rows are checked by discrete logarithm.
            h_rows, h_len, h_bits, hk0, hd = 4096, 8192, 3, 0xFEED, 3
            if h_rows % world == 0:
                rlo, rhi = shard_bounds(h_rows, world, rank)
                doc = np.random.default_rng(0xD0C).integers(0, 7, size=(h_rows, h_len), dtype=np.uint8)     # the same document on every rank
                d_sym = msm.DeviceBuffer.from_host(np.ascontiguousarray(doc[rlo:rhi]))
                hctx = msm.MsmContext(a.curve, msm.gen_bases(a.curve, hk0, hd, h_len, device=True), h_len)
                mine = torch.zeros(96 * (rhi - rlo), dtype=torch.uint8, device=dev)
                allr = torch.zeros(96 * h_rows, dtype=torch.uint8, device=dev)
                hx = make_exch(hctx)

                def hyrax_once():
                    hctx.msm_rows_symbols(d_sym, rhi - rlo, h_len, h_bits, out=mine.data_ptr())
                    hx.all_gather(mine, allr)
                for _ in range(2):
                    hyrax_once()
                hctx.sync()
                torch.cuda.synchronize()
                dist.barrier()
                t0_ = time.perf_counter()
                for _ in range(ksteps):
                    hyrax_once()
                hctx.sync()
                torch.cuda.synchronize()
                dist.barrier()
                t_ = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64, device=cdev0)
                dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                got = allr.cpu().numpy().reshape(h_rows, 96)
                from oracle.pasta_oracle import CURVES
                order = CURVES[a.curve].order
                rows_ok = True
                for r_ in (0, h_rows // world - 1, h_rows // 2, h_rows - 1):          # rows of several ranks
                    dl = sum(int(v) * (hk0 + j * hd) for j, v in enumerate(doc[r_].tolist())) % order
                    rows_ok = rows_ok and msm.compress(a.curve, got[r_].copy().view(np.uint64)) == point_of_dlog(a.curve, dl)
                strong["hyrax_rows"] = {"rows": h_rows, "row_len": h_len, "symbol_bits": h_bits, "ms_per_commit": float(t_.item()) / ksteps * 1e3,
                                        "rows_per_rank": rhi - rlo, "check": "dlog-ok" if rows_ok else "MISMATCH",
                                        "note": "HyraxPC::commit of a 16 MiB DNA document, rows sharded in contiguous blocks, all-gather of the row commitments"}
                strong_results["hyrax_rows_ok"] = rows_ok
                hctx.close()
                d_sym.free()

        except Exception as e:
            print(f"[bench] strong-scaling legs failed on rank {rank}: {e}", file=sys.stderr)
            strong, strong_results = {"error": str(e)}, {}
        # (round 4) the work Reef's cfg4 really does, placed over the ranks as WHOLE units: the final SNARK's three
        # arguments (src/backend/framework.rs:695-721) and one folding step's sum-check with its table sharded by low index bits
        # (r1cs_helper.rs:441-544).  Their tiny exchanges feed the HOST's transcript, so they go through a gloo group.
        if isinstance(strong, dict) and "error" not in strong and os.environ.get("REEF_BENCH_UNITS", "1") != "0":
            try:
                strong.update(independent_units_legs(a, rank, world, dist))
            except Exception as e:
                strong["independent_units_error"] = str(e)
        # (round 5) what ONE process gets from the same N devices through the library's device groups (reef_msm_group_*: the form a Rust prover can
        # call; Reef is one process, src/backend/main.rs:82) -- a child of rank 0 with no launcher variables runs `bench.py --gpus N --single-process`
        # over every visible device while the other ranks wait; a child, so that a first-time failure of the untested cross-device calls (peer
        # copies, cross-device event waits) costs this field, not the line.
        if isinstance(strong, dict) and "error" not in strong and os.environ.get("REEF_BENCH_SINGLE", "1") != "0":
            if rank == 0:
                try:
                    import subprocess
                    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                                                                          "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE", "TORCHELASTIC_RUN_ID") and not k.startswith("TORCHELASTIC")}
                    for field, ex in (("single_process", "peer"), ("single_process_rccl", "rccl")):
                        child = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", str(a.gpus), "--single-process", "--group-exchange", ex, "--logn", str(a.logn),
                                                "--steps", "2", "--warmup", "1", "--msms-per-step", "12", "--curve", a.curve], capture_output=True, text=True, timeout=150, env=env)
                        lines = [ln for ln in child.stdout.splitlines() if ln.startswith("{")]
                        if child.returncode == 0 and lines:
                            cl = json.loads(lines[-1])
                            cc = cl["config"]
                            strong[field] = {"value": cl["value"], "ms_per_msm": cc["ms_per_msm"], "host_scalars_ms_per_msm": cc["host_scalars_ms_per_msm"],
                                             "devices": cc["devices"], "distinct_devices": cc["distinct_devices"], "exchange": cc["exchange"], "peer_members": cc["peer_members"],
                                             "check": cc["check"], "strong_scaling": cc["strong_scaling"], "devices_note": cc["devices_note"],
                                             "note": "`bench.py --gpus N --single-process` run by rank 0 as a child while the other ranks wait: one process, N members, "
                                                     "the library's own exchange (no torch.distributed); value = pairs/s of one N x 2^logn-point MSM per call, weak scaling by points"}
                            if ex == "rccl":                      # the same protocol, the 96-byte partial sums as ncclSend/ncclRecv on a single-process communicator
                                strong[field].pop("note")
                                strong[field].pop("devices")
                        else:
                            strong[field] = {"error": f"child exited with {child.returncode}: {child.stderr[-400:]}"}
                            break
                except Exception as e:
                    strong["single_process"] = {"error": str(e)}
            dist.barrier()

    # ---- after the timed region (none of this is `value`) ------------------------------------------------
    # (1) the accumulation kernel with ONE MSM in flight: with several MSMs sharing the chip a launch is stretched by
    #     its neighbours, so the per-launch duration above understates the kernel
    if not multi:
        reps = max(3, min(10, a.steps * MPS))
        for _ in range(reps):
            ctx0.msm(scalars, n, out=results[0].data_ptr())
            ctx0.sync()
        s1 = ctx0.timing_stats(reset=True)
        single = {"kernel_ms": s1["accumulate_ms"] / max(s1["calls"], 1), "msm_ms": s1["total_ms"] / max(s1["calls"], 1)}
        # (1b) the VARIABLE-BASE figure: the same points and scalars on a key with NO precomputed tables (bucket_groups = 0:
        #      one bucket set per window, nothing but the imported points resident) -- pasta-msm's contract, and what the zero-patch drop-in gets
        #      for bases it sees once (the IPA's folded generators, framework.rs:695-698).  One MSM in flight, then `streams` in flight.
        try:
            torch.cuda.synchronize()
            t_pk = time.perf_counter()
            pk0 = msm.MsmContext(a.curve, bases, n, bucket_groups=0)
            plain_build_ms = (time.perf_counter() - t_pk) * 1e3
            pks = [pk0] + [pk0.clone() for _ in range(nctx - 1)]
            pres = [torch.zeros(96, dtype=torch.uint8, device=dev) for _ in pks]
            for j, c in enumerate(pks):                       # warm-up: workspaces
                c.msm(scalars, n, out=pres[j].data_ptr())
                c.sync()
            t_pk = time.perf_counter()
            for _ in range(reps):
                pk0.msm(scalars, n, out=pres[0].data_ptr())
                pk0.sync()
            plain_one_ms = (time.perf_counter() - t_pk) / reps * 1e3
            kk = max(12, min(60, a.steps * MPS))
            t_pk = time.perf_counter()
            for i in range(kk):
                pks[i % nctx].msm(scalars, n, out=pres[i % nctx].data_ptr())
            for c in pks:
                c.sync()
            plain_flight_ms = (time.perf_counter() - t_pk) / kk * 1e3
            pplan = pk0.plan()
            same = a.no_check or msm.compress(a.curve, pres[0].cpu().numpy().view(np.uint64)) == msm.compress(a.curve, final_result[:96].view(np.uint64))
            plain_key = {"ms_per_msm": plain_flight_ms, "pairs_per_s": n / (plain_flight_ms * 1e-3), "one_in_flight_ms_per_msm": plain_one_ms,
                         "in_flight": nctx, "window_bits": pplan["window_bits"], "windows": pplan["windows"], "tables": pplan["tables"],
                         "key_build_ms": plain_build_ms, "check": "skipped" if a.no_check else ("same-point-as-the-fixed-key-MSM" if same else "MISMATCH"),
                         "note": "no precompute: the key is the imported points only (64 B per point resident); device-resident scalars and result, like `value`"}
            if not same:
                raise RuntimeError("plain-key MSM differs from the fixed-key MSM")
            for c in pks:
                c.close()
        except Exception as e:
            print(f"[bench] plain-key leg failed: {e}", file=sys.stderr)
            plain_key = {"error": str(e)}
            if "differs" in str(e):
                raise
        # (2) the same workload with the scalars in pinned HOST memory and the result returned to the host (what Reef's
        #     prover hands over): PCIe-inclusive, the same number of MSMs in flight on their own streams
        try:
            pin = torch.empty((n, 4), dtype=torch.int64, pin_memory=True)
            pin.copy_(torch.from_numpy(scalars.to_host((n, 4)).view(np.int64)))
            hs = pin.numpy().view(np.uint64)
            import threading
            # six caller threads, each on its own clone of the resident key: uploads (0.62 ms per 32 MiB at 54 GB/s), sorts and
            # tails of some calls run under the accumulation of others (tools/time_host_scalars.py: 1 / 3 / 6 threads)
            hthreads = max(nctx, 6)
            hctx = list(ctxs) + [ctxs[0].clone() for _ in range(hthreads - nctx)]
            outs = [np.zeros(12, dtype=np.uint64) for _ in hctx]
            per_thread = max(8, min(a.steps * MPS, 48) // hthreads + 1)

            def run_host(buf, nthreads):      # one caller thread per resident-key clone, as nova's rayon workers would be; -> ms per MSM
                def host_worker(j):
                    for _ in range(per_thread):
                        hctx[j].msm(buf, n, out=outs[j])
                for j in range(nthreads):     # warm-up (staging buffers)
                    hctx[j].msm(buf, n, out=outs[j])
                th = [threading.Thread(target=host_worker, args=(j,)) for j in range(nthreads)]
                t1 = time.perf_counter()
                [x.start() for x in th]
                [x.join() for x in th]
                return (time.perf_counter() - t1) / (per_thread * nthreads) * 1e3
            host_ms = run_host(hs, hthreads)
            # the same from PAGEABLE memory -- what a Rust Vec is: the runtime stages or pins it per call
            hp = np.array(hs, copy=True)
            host_pageable = {"one_caller": run_host(hp, 1), "six_callers": run_host(hp, hthreads)}
            for c in hctx[nctx:]:
                c.close()
            if not a.no_check and msm.compress(a.curve, outs[0]) != msm.compress(a.curve, final_result.view(np.uint64)):
                raise RuntimeError("host-scalar MSM differs from the device-scalar MSM")
        except Exception as e:                 # never let the side measurement take the bench line down
            print(f"[bench] host-scalar timing skipped: {e}", file=sys.stderr)

    _diag("after the host-scalar leg")
    check = "skipped"
    partials_differ = None
    if not a.no_check:
        # size-independent parity check of the last result (outside the timed region): bases are an arithmetic
        # progression, so every MSM over them has a known discrete log
        if not multi:
            canon = msm.gen_scalars(a.curve, seed, n * B, kind=kind, mont=False)
            my_dlog = dlog_of_msm(a.curve, canon[:n], k0, d, owner * n)
            ok = msm.compress(a.curve, final_result[:96].view(np.uint64)) == point_of_dlog(a.curve, my_dlog)
            for r in range(1, B):             # every MSM of the batch
                ok = ok and msm.compress(a.curve, final_result[96 * r:96 * (r + 1)].view(np.uint64)) == point_of_dlog(
                    a.curve, dlog_of_msm(a.curve, canon[r * n:(r + 1) * n], k0, d, 0))
        else:
            # the timed region's own points were checked before the side legs (weak_check above); what is left are the strong legs' results:
            # both splits computed rank 0's MSM, whose discrete logarithm is the first gathered one
            ok, partials_differ, dlog0 = weak
            if strong_results:
                want0 = point_of_dlog(a.curve, dlog0)
                for name_, res_ in strong_results.items():
                    if isinstance(res_, bool):
                        ok = ok and res_
                    else:
                        ok = ok and msm.compress(a.curve, res_.view(np.uint64)) == want0
            flag = torch.tensor([1 if ok else 0], device=dev if a.backend == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag[0].item())
        check = "dlog-ok" if ok else "MISMATCH"

    if rank == 0:
        out = build_line(check, partials_differ, strong)
        side_legs = a.gpus == 1 and not multi and not a.no_replay
        if side_legs:                      # GPU side legs first: they are host-latency sensitive (see replay_leg)
            _diag("before closing the contexts")
            for c_ in ctxs:                # the timed region's contexts are done: their streams would share hardware queues with the legs'
                try:
                    c_.close()
                except Exception:
                    pass
            _diag("contexts closed")
            try:                           # the HBM-bound row of the path (N2) beside the issue-bound headline kernel
                out["roofline"]["hbm_bound_row"] = hbm_bound_leg()
            except Exception as e:
                out["roofline"]["hbm_bound_row"] = {"error": str(e)}
            _diag("after the HBM-bound leg")
            for cfg_ in ("cfg3", "cfg4"):  # BASELINE configs[2] and configs[3] (north_star's own target document: 16 MiB)
                try:
                    out["config"]["replay_" + cfg_] = replay_leg(cfg_, with_tables=cfg_ == "cfg3")
                except Exception as e:     # a side measurement never takes the bench line down
                    out["config"]["replay_" + cfg_] = {"error": str(e)}
        if a.gpus == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(msm.curve_id(a.curve), a.cpu_seconds)
            for cfg_ in ("cfg3", "cfg4"):
                if side_legs and "error" not in out["config"]["replay_" + cfg_]:
                    try:
                        replay_cpu_leg(out["config"]["replay_" + cfg_], cfg_, cpu_threads=out["cpu_baseline"].get("cores"))
                    except Exception as e:
                        out["config"]["replay_" + cfg_]["cpu_error"] = str(e)
        emit(out)
    if multi:
        dist.barrier()
    run_finished.set()
    if multi:
        dist.destroy_process_group()
    if check == "MISMATCH":
        sys.exit(2)


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        if WATCHDOG_FIRED.is_set():    # the side legs' deadline has passed and this rank's collective broke because a peer left: the watchdog ends the process
            time.sleep(60)
        raise
