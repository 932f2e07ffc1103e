#!/usr/bin/env python3
"""Generate tests/golden/*.json from the big-int oracle (oracle/pasta_oracle.py).

Run from the repo root:  python oracle/gen_golden.py
TEST INFRASTRUCTURE ONLY.  The reference holds no MSM known-answer vectors
(SURVEY.md 8c), so these are produced by the oracle's textbook affine group law and are
cross-checked here two ways before being written: naive double-and-add vs the bucket
method, and (for arithmetic-progression bases) vs the discrete-log closed form.
"""
from __future__ import annotations

import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.pasta_oracle import (CURVES, SplitMix64, ap_bases, msm_via_dlog, sha_hex,  # noqa: E402
                                 uniform_scalar)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def witness_like(rng: SplitMix64, order: int) -> int:
    """Mirror of pasta_ref_gen_scalars kind=1 (4 draws per scalar; class from top byte of w[3])."""
    w = [rng.next() for _ in range(4)]
    sel = (w[3] >> 56) % 10
    if sel < 7:
        return w[0] & 1
    if sel < 9:
        return w[0] & 0xFFFF
    v = (w[0] | (w[1] << 64) | (w[2] << 128) | (w[3] << 192)) & ((1 << 255) - 1)
    return v - order if v >= order else v


def small(rng: SplitMix64, bound: int) -> int:
    w = [rng.next() for _ in range(4)]
    return w[0] % bound


def anchors():
    out = []
    for name, C in CURVES.items():
        G = C.gen
        for k in (1, 2, 3, 5, 100, C.order - 1):
            pt = C.mul(k, G)
            out.append({"curve": name, "k": hex(k), "x": hex(pt[0]), "y": hex(pt[1]),
                        "compressed": C.compress(pt).hex()})
    return out


def seeded_cases():
    cases = []
    sizes = [1, 2, 3, 127, 128, 129, 1000, 4096]
    for name, C in CURVES.items():
        for n in sizes:
            for kind in (0, 1):
                if kind == 1 and n not in (128, 1000, 4096):
                    continue
                seed = 0x5EEF + n + 1000 * kind
                k0, d = 7 + n, 3
                rng = SplitMix64(seed)
                if kind == 0:
                    sc = [uniform_scalar(rng, C.order) for _ in range(n)]
                else:
                    sc = [witness_like(rng, C.order) for _ in range(n)]
                bases = ap_bases(C, k0, d, n)
                res = C.msm(sc, bases, c=8)
                assert res == msm_via_dlog(C, sc, k0, d), (name, n, kind)
                if n <= 129:
                    assert res == C.msm_naive(sc, bases), (name, n, kind)
                inp = b"".join(C.affine_to_bytes(b) for b in bases) + b"".join(C.scalar_to_bytes(s) for s in sc)
                cases.append({"curve": name, "n": n, "seed": seed, "kind": kind, "k0": k0, "d": d,
                              "input_sha256": sha_hex(inp),
                              "expect_compressed": C.compress(res).hex(),
                              "expect_x": hex(res[0]) if res else "inf",
                              "expect_y": hex(res[1]) if res else "inf"})
                print("seeded", name, n, kind, file=sys.stderr)
    return cases


def explicit_cases():
    """Edge cases with every input spelled out (hex of the C-ABI byte layout)."""
    cases = []
    for name, C in CURVES.items():
        G = C.gen
        q = C.order
        P5, P9 = C.mul(5, G), C.mul(9, G)

        def add_case(label, scalars, bases):
            res = C.msm_naive(scalars, bases)
            assert res == C.msm(scalars, bases, c=5), label
            cases.append({
                "curve": name, "label": label,
                "scalars_mont_hex": [C.scalar_to_bytes(s, mont=True).hex() for s in scalars],
                "scalars_canon_hex": [C.scalar_to_bytes(s, mont=False).hex() for s in scalars],
                "bases_hex": [C.affine_to_bytes(b).hex() for b in bases],
                "expect_compressed": C.compress(res).hex()})

        add_case("kat_100G", [(i + 1) ** 2 for i in range(4)], [C.mul(i + 1, G) for i in range(4)])
        add_case("empty", [], [])
        add_case("all_zero_scalars", [0, 0, 0], [G, P5, P9])
        add_case("scalar_q_minus_1", [q - 1], [P5])
        add_case("scalar_q_minus_1_and_one", [q - 1, 1], [P5, P5])          # cancels to identity
        add_case("duplicate_bases", [3, 4, 5, 6], [P5, P5, P5, P5])
        add_case("duplicate_bases_same_scalar", [7, 7, 7], [P9, P9, P9])    # forces P+P in a bucket
        add_case("p_and_minus_p", [11, 11], [P5, C.neg(P5)])                # identity
        add_case("identity_base", [3, 99, 4], [G, None, P9])
        add_case("only_identity_bases", [3, 99], [None, None])
        add_case("all_equal_small", [4] * 9, [C.mul(i + 2, G) for i in range(9)])
        add_case("powers_of_two", [1 << (16 * i) for i in range(16)], [C.mul(i + 3, G) for i in range(16)])
        add_case("window_edges", [(1 << 255) % q, q - 2, (1 << 254) - 1, 0x8000, 0x7FFF, 0xFFFF, 0x10000],
                 [C.mul(i + 20, G) for i in range(7)])
        add_case("max_digit_run", [int("8" * 63, 16) % q, int("f" * 63, 16) % q], [P5, P9])
    return cases


def fold_cases():
    cases = []
    for name, C in CURVES.items():
        rng = SplitMix64(0xF01D)
        gens = ap_bases(C, 31, 5, 8)
        w1, w2 = uniform_scalar(rng, C.order), uniform_scalar(rng, C.order)
        out = [C.add(C.mul(w1, gens[i]), C.mul(w2, gens[4 + i])) for i in range(4)]
        cases.append({"curve": name, "k0": 31, "d": 5, "n": 8, "w1": hex(w1), "w2": hex(w2),
                      "expect_compressed": [C.compress(o).hex() for o in out]})
        # degenerate: w1 = 0, w2 = 1 and L_i == R_i with w1 = w2 (doubling inside the fold)
        out = [gens[4 + i] for i in range(4)]
        cases.append({"curve": name, "k0": 31, "d": 5, "n": 8, "w1": hex(0), "w2": hex(1),
                      "expect_compressed": [C.compress(o).hex() for o in out]})
    return cases


def row_cases():
    """Hyrax-style row commitments: L rows over the same R bases, small symbols + blind*H."""
    cases = []
    for name, C in CURVES.items():
        rows, row_len, bound = 4, 16, 7      # DNA-like symbols 0..6 (framework.rs:978-1011)
        rng = SplitMix64(0xD0C)
        sc = [small(rng, bound) for _ in range(rows * row_len)]
        blinds = [uniform_scalar(rng, C.order) for _ in range(rows)]
        bases = ap_bases(C, 41, 2, row_len)
        H = C.mul(0xB11D, C.gen)
        outs = []
        for r in range(rows):
            pt = C.msm_naive(sc[r * row_len:(r + 1) * row_len], bases)
            pt = C.add(pt, C.mul(blinds[r], H))
            outs.append(C.compress(pt).hex())
        cases.append({"curve": name, "rows": rows, "row_len": row_len, "bound": bound, "seed": 0xD0C,
                      "k0": 41, "d": 2, "h_k": 0xB11D,
                      "scalars": sc, "blinds": [hex(b) for b in blinds], "expect_compressed": outs})
    return cases


def main():
    os.makedirs(OUT, exist_ok=True)
    data = {
        "_comment": "Generated by oracle/gen_golden.py from oracle/pasta_oracle.py (big-int definition of "
                    "Pallas/Vesta). The reference holds no MSM KATs (SURVEY.md 8c): parity unpinned by the "
                    "reference; anchors 2G/3G/5G/(q-1)G/100G equal the values listed in SURVEY.md 8c.",
        "anchors": anchors(),
        "explicit": explicit_cases(),
        "seeded": seeded_cases(),
        "fold": fold_cases(),
        "rows": row_cases(),
    }
    with open(os.path.join(OUT, "pasta_msm_golden.json"), "w") as f:
        json.dump(data, f, indent=1)
    print("wrote", os.path.join(OUT, "pasta_msm_golden.json"))


if __name__ == "__main__":
    main()
