"""The MSM sequence of one `reef --prove` run on the HOST cores, through oracle/pasta_ref.c.  TEST INFRASTRUCTURE ONLY:
the `cpu_restatement` leg that bench.py reports beside the GPU replay of the same sequence (never the product path).

The sequence is the one reef_amd/csrc/host/reef_replay.cpp issues (eniac/Reef src/backend/framework.rs:664-723), with the
shapes of tests/golden/replay_shapes.json:
  per folding step   comm_T2, comm_W1, comm_T1, comm_W2: four MSMs (window-parallel Pippenger on the thread pool)
  final SNARK        one more |C2| MSM, then per curve an inner-product argument over the padded key: log2 N rounds of two
                     cross-term MSMs of half the generators + the generator fold G' = w1 G_lo + w2 G_hi (one joint
                     double-and-add per pair, dealt out to the pool) -- what nova's ipa_pc does on the CPU [recalled]
  consistency        the same argument over the Hyrax row generators (src/backend/commitment.rs:371,383)
It is a CPU RESTATEMENT, not the reference binary (Reef is Rust; cargo is absent here); the thread count is reported.
"""
from __future__ import annotations

import json
import os
import time

from . import pasta_ref as R


def _next_pow2(x: int) -> int:
    p = 1
    while p < x:
        p <<= 1
    return p


def load_shape(config: str, shapes_path: str) -> dict:
    with open(shapes_path) as f:
        for s in json.load(f)["shapes"]:
            if config in s["name"]:
                return s
    raise KeyError(config)


def _ipa(curve: int, gens, scalars, n: int, threads: int) -> float:
    t0 = time.perf_counter()
    cur = gens[:n]
    w1 = 0x03333333444444441111111122222222_0FEDCBA987654321_1234567890ABCDEF   # fixed stand-ins for the fold challenges
    w2 = 0x07777777888888885555555566666666_0123456789ABCDEF_0BADC0FFEE0DDF00
    ln = n
    while ln > 1:
        half = ln // 2
        lo, hi = cur[:half], cur[half:ln]
        R.msm_pippenger_windows(curve, hi, scalars[:half], threads=threads)          # L = <a_lo, G_hi>
        R.msm_pippenger_windows(curve, lo, scalars[half:ln], threads=threads)        # R = <a_hi, G_lo>
        cur = R.fold_mt(curve, cur, w1, w2, threads, half=half)             # G' = w1 G_lo + w2 G_hi
        ln = half
    return (time.perf_counter() - t0) * 1e3


def run(config: str, shapes_path: str, threads: int) -> dict:
    sh = load_shape(config, shapes_path)
    n1 = _next_pow2(max(sh["w1"], sh["c1"]))
    n2 = _next_pow2(max(sh["w2"], sh["c2"]))
    g1 = R.gen_bases_ap(R.PALLAS, 0xC0FFEE, 7, n1)
    g2 = R.gen_bases_ap(R.VESTA, 0xC0FFEE + 1, 7, n2)
    sW1, sT1 = R.gen_scalars(R.PALLAS, 11, n1, kind=1), R.gen_scalars(R.PALLAS, 12, n1, kind=0)
    sW2, sT2 = R.gen_scalars(R.VESTA, 13, n2, kind=1), R.gen_scalars(R.VESTA, 14, n2, kind=0)

    def msm(curve, gens, sc, n):
        return R.msm_pippenger_windows(curve, gens, sc, threads=threads, n=n)

    msm(R.PALLAS, g1, sW1, sh["w1"])                                                 # warm-up: the pool's threads exist from here on
    t0 = time.perf_counter()
    for _ in range(sh["steps"]):
        msm(R.VESTA, g2, sT2, sh["c2"])
        msm(R.PALLAS, g1, sW1, sh["w1"])
        msm(R.PALLAS, g1, sT1, sh["c1"])
        msm(R.VESTA, g2, sW2, sh["w2"])
    steps_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    msm(R.VESTA, g2, sT2, sh["c2"])
    ipa1 = _ipa(R.PALLAS, g1, sT1, n1, threads)
    ipa2 = _ipa(R.VESTA, g2, sT2, n2, threads)
    final_ms = (time.perf_counter() - t0) * 1e3
    cons_ms = _ipa(R.PALLAS, g1, sT1, sh["hyrax_row"], threads) if sh["hyrax_row"] >= 2 else 0.0
    return {"fold_steps_ms": steps_ms, "ms_per_step": steps_ms / sh["steps"], "ipa_pallas_ms": ipa1, "ipa_vesta_ms": ipa2, "final_snark_ms": final_ms,
            "consistency_ipa_ms": cons_ms, "total_prove_msm_ms": steps_ms + final_ms + cons_ms, "threads": threads,
            "kind": "port (oracle/pasta_ref.c: window-parallel Pippenger + joint double-and-add generator folds on a thread pool; NOT the reference binary)"}
