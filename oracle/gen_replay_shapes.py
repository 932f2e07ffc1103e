#!/usr/bin/env python3
"""Evaluates Reef's constraint-count model (oracle/costs_oracle.py = src/backend/costs.rs restated) for the configs of
BASELINE.json and writes tests/golden/replay_shapes.json: the per-config MSM lengths the replay harness
(reef_amd/csrc/host/reef_replay.cpp) issues.  TEST INFRASTRUCTURE ONLY.

    python oracle/gen_replay_shapes.py            # rewrites tests/golden/replay_shapes.json

What is derived: the step-circuit size (full_round_cost_model), |W1| = |C1| = V1 + cost, |W2| = |C2| = V2 (the
two terms of get_folded_cost, costs.rs:168-179), the number of folding steps (ceil(trace / batch)), the document
sizes (padded document: bytes + EOF + EPSILON rounded up to a power of two, src/backend/framework.rs:997-1008;
hybrid table 2 * max(table, document), src/backend/r1cs.rs:481-487; Hyrax matrix 2^(l/2) x 2^(l - l/2),
src/backend/commitment.rs:173-174).

What is an INPUT: the regex, the charset and the flags of each config -- taken from the reference's own scripts and README where it
holds them (file:line below) -- and the document length BASELINE.json names.  The automaton's shape is no longer typed in (rounds 1-4
assumed "one edge per literal character"): oracle/safa_shape.py restates SAFA::new for the regex family of those scripts (skips and
literals) and gives num_states, num_edges, the largest skip offset and the solution length the cost model is fed.  Nothing is typed
into the harness.
"""
from __future__ import annotations

import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import costs_oracle as K  # noqa: E402
from oracle import safa_shape as S  # noqa: E402

# charsets: src/config.rs:229-233 (ascii: 128 characters), :250-253 (utf8: every Unicode scalar value), :266-268 (dna: ACGT)
ALPHABETS = {"ascii": (128, None), "dna": (4, "ACGT"), "utf8": (0x110000 - 0x800, None)}
# The BRCA regexes of tests/scripts/dna.sh address a 10 000-base gene region appended to a long base document: `^.{k}` skips the base
# and the region's prefix (k = |base| + offset; in the 1 MB documents the tree holds, tests/docs/BRCA1_base1m+primary and
# BRCA2_base1m+primary, the literals sit at 1 000 000 + 8129 / 5784 / 1970).  For a document of N bytes the same regex is taken with
# k = N - 10 000 + offset: the only number that changes with the document, as in the reference's own 1 MB / full-size pairs.
BRCA1_A = "ATGGGCTACAGAAACCGTGCCAAAAGACTTCTACAGAGTGAACCCGAAAATCCTTCCTTG"
BRCA1_B = ("ATGCTGAAACTTCTCAACCAGAAGAAAGGGCCTTCACAGTGTCCTTTATGTAAGAATGATATAACCAAAAG", "AGCCTACAAGAAAGTACGAGATTTAGTCAACTTGTTGAAGAGCTATTGAAAATCATTTGTGCTTTTCAGCTTGACACAGGTTTGGAGT",
           "ATGCAAACAGCTATAATTTTGCAAAAAAGGAAAATAACTCTCCTGAACATCTAAAAGATGAAGTTTCTATCATCCAAAGTATGGGCTACAGAAACCGTGCCAAAAGACTTCTACAGAGTGAACCCGAAAATCCTTCCTTG")
BRCA2_LENS = (428, 182, 188, 171)          # the four literals of dna.sh:12-13 (lengths; any ACGT text of these lengths gives the same automaton shape)


def brca(doc_bytes: int, offset: int, literals) -> str:
    return "^.{%d}" % (doc_bytes - 10000 + offset) + ".*".join(literals)


CONFIGS = [
    dict(name="cfg1_9B_ascii", regex=".*b", source="README.md:63 (`reef --input document --re '.*b' ... ascii`, BASELINE configs[0])", charset="ascii",
         doc_bytes=9, alphabet_bits=8, hybrid=False, merkle=False, batch=0),
    dict(name="cfg3_1MiB_ascii_password", regex=".*password.*", source="BASELINE.json configs[2] (the reference's password scripts use lookaheads: tests/scripts/password.sh; this "
         "regex is BASELINE's own)", charset="ascii", doc_bytes=1 << 20, alphabet_bits=8, hybrid=False, merkle=False, batch=0),
    dict(name="cfg4_16MiB_dna_hybrid_b32", regex=brca(1 << 24, 8129, [BRCA1_A]), source="tests/scripts/dna.sh:8,19 (the regex of :6 on a matching document): `^.{k}` + the 60-base BRCA1 "
         "literal, k re-based to a 16 MiB document", charset="dna", doc_bytes=1 << 24, alphabet_bits=3, hybrid=True, merkle=False, batch=32,
         hand_count="nodes: root + 60 literal states + `.*` + empty suffix + sink = 64; edges: skip + complement (2) + sink loop (1) + 60 x (epsilon + ACGT) + `.*` (1) + "
                    "empty suffix (5) = 309; longest accepting path 62 edges -> solution length 63 -> ceil(63 / 32) = 2 folding steps (tests/test_safa_shape.py)"),
    dict(name="cfg4b_16MiB_dna_three_literals_hybrid_b32", regex=brca(1 << 24, 5784, BRCA1_B), source="tests/scripts/dna.sh:11,22 (the regex of :7): three BRCA1 literals (71, 88, 140 bases) "
         "joined by `.*`", charset="dna", doc_bytes=1 << 24, alphabet_bits=3, hybrid=True, merkle=False, batch=32,
         hand_count="nodes: root + 299 literal states + three `.*` + empty suffix + sink = 305; edges: 2 + 1 + 299 x 5 + 3 + 5 = 1506; solution length 304 -> 10 folding steps"),
    dict(name="cfg5_64MiB_utf8_merkle", regex=brca(1 << 26, 1970, ["A" * n for n in BRCA2_LENS]), source="tests/scripts/dna.sh:13,24 (the regex of :12): four BRCA2 literals (428, 182, 188, 171 "
         "bases) joined by `.*`, on a document read with the utf8 charset as BASELINE configs[4] names it (--merkle excludes projections and --hybrid: r1cs.rs:511-512)",
         charset="utf8", doc_bytes=1 << 26, alphabet_bits=8, hybrid=False, merkle=True, batch=0),
]


def safa_of(cfg: dict) -> S.Shape:
    size, chars = ALPHABETS[cfg["charset"]]
    return S.shape(cfg["regex"], size, chars)


def evaluate(cfg: dict) -> dict:
    udoc_len = K.next_power_of_two(cfg["doc_bytes"] + 2)                 # + EOF + EPSILON, zero-padded (framework.rs:997-1008)
    doc_log = K.logmn(udoc_len)
    sh = safa_of(cfg)
    trace = list(sh.path_lens)                                           # `final_paths`: what NFA::new hands the cost model (r1cs.rs:335, :496-506)
    safa = K.SafaShape(num_states=sh.num_states, num_edges=sh.num_edges, max_offset=sh.max_offsets, max_branches=1, max_stack=1)
    table = K.next_power_of_two(safa.num_edges)
    hybrid_len = 2 * K.next_power_of_two(max(table, udoc_len)) if cfg["hybrid"] else None   # r1cs.rs:481-487
    batch = cfg["batch"] or max(2, K.opt_cost_model_select(safa, udoc_len, cfg["hybrid"], hybrid_len, False, trace))   # r1cs.rs:489-513 (> 1)
    if cfg["merkle"]:
        # costs.rs has no Merkle term: the document lookups of nl_doc are replaced by b Merkle paths (nova.rs:392-511).  Rounds 1-5 extrapolated b * log2 N * 288;
        # round 6 counts the gadget's rows by hand (K.merkle_gadget: selects, one width-5 permutation per hash, ensure_allocated, the root equality).  The batch size is
        # still chosen by costs.rs' model WITHOUT the Merkle term (that is what Reef does: opt_cost_model_select knows nothing of --merkle).
        step = (K.nl(batch, table, False) + K.lookup_idxs(safa.num_states, batch) + K.cursor_circuit(udoc_len, batch, safa.max_offset)
                + K.stack_circuit(safa.num_states, udoc_len, safa.max_branches, safa.max_stack) + K.merkle_gadget(batch, udoc_len))
    else:
        step = K.full_round_cost_model(safa, batch, udoc_len, cfg["hybrid"], hybrid_len, False)
    steps = K.n_foldings(trace, batch)
    primary = K.V1 + step
    table_log = K.logmn(hybrid_len) if cfg["hybrid"] else doc_log
    out = dict(name=cfg["name"], w1=primary, c1=primary, w2=K.V2, c2=K.V2, steps=steps, batch=batch, step_circuit_constraints=step,
               hyrax_row=0 if cfg["merkle"] else 1 << (doc_log - doc_log // 2), doc_log=0 if cfg["merkle"] else doc_log,
               symbol_bits=cfg["alphabet_bits"], table_log=0 if cfg["merkle"] else table_log, lookups=2 * batch if cfg["hybrid"] else batch,
               merkle_log=doc_log if cfg["merkle"] else 0,
               folded_cost=K.get_folded_cost(step, trace, batch),
               safa_states=sh.num_states, safa_edges=sh.num_edges, safa_max_offsets=sh.max_offsets, solution_lens=trace)
    if cfg["merkle"]:
        out["merkle_gadget_constraints"] = K.merkle_gadget(batch, udoc_len)
        out["merkle_gadget_basis"] = ("hand count of NFAStepCircuit::eval_merkle (src/backend/nova.rs:392-511): per lookup 4 + 289 rows for the leaf, 2 + 289 per inner level "
                                      f"({doc_log - 1} of them), 1 for the root equality = {K.merkle_gadget(1, udoc_len)} rows; costs.rs has no such term")
    return out


def main() -> None:
    shapes = [evaluate(c) for c in CONFIGS]
    doc = {
        "generated_by": "oracle/gen_replay_shapes.py (oracle/costs_oracle.py restates src/backend/costs.rs of eniac/Reef)",
        "constants": {"V1": K.V1, "V2": K.V2},
        "inputs": [{k: (v if k != "regex" or len(v) < 200 else v[:120] + "..." + v[-40:]) for k, v in c.items()} for c in CONFIGS],
        "note": "MSM lengths are PREDICTIONS of Reef's cost model (src/backend/costs.rs restated) for automata DERIVED from the regexes above by "
                "oracle/safa_shape.py (SAFA::new restated for skips and literals), not measurements of a Reef run",
        "shapes": shapes,
    }
    path = os.path.join(ROOT, "tests", "golden", "replay_shapes.json")
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    for s in shapes:
        print(s)


if __name__ == "__main__":
    main()
