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

What is an INPUT (Reef's frontend is not restated, and the reference records no SAFA sizes for these documents): the SAFA
shape and the trace length of each regex, given below with the reasoning.  Change them here and re-run; nothing is typed
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

# ---- inputs: SAFA shape and solver trace per config (assumptions, see the module docstring) ------------------------------
#  states/edges: one state per literal character of the regex plus the skip/accept states (safa.rs:86-209 builds one
#  node per derivative); `.*` and `.{n}` are ONE skip edge (safa.rs:600), every literal character is one transition
#  (safa.rs:364-368).  max_offset = document length for an unbounded skip.  No alternation/lookahead: one branch, stack 1.
CONFIGS = [
    dict(name="cfg1_9B_ascii", regex=".*b", doc_bytes=9, alphabet_bits=8, hybrid=False, merkle=False, batch=0,
         safa=dict(num_states=4, num_edges=4, max_branches=1, max_stack=1), trace=[3]),
    dict(name="cfg3_1MiB_ascii_password", regex=".*password.*", doc_bytes=1 << 20, alphabet_bits=8, hybrid=False, merkle=False, batch=0,
         safa=dict(num_states=12, num_edges=12, max_branches=1, max_stack=1), trace=[11]),
    dict(name="cfg4_16MiB_dna_hybrid_b32", regex="DNA motif, ~128 transitions", doc_bytes=1 << 24, alphabet_bits=3, hybrid=True, merkle=False, batch=32,
         safa=dict(num_states=130, num_edges=130, max_branches=1, max_stack=1), trace=[128]),
    dict(name="cfg5_64MiB_utf8_merkle", regex="literal match, ~128 transitions", doc_bytes=1 << 26, alphabet_bits=8, hybrid=False, merkle=True, batch=0,
         safa=dict(num_states=130, num_edges=130, max_branches=1, max_stack=1), trace=[128]),
]


def evaluate(cfg: dict) -> dict:
    udoc_len = K.next_power_of_two(cfg["doc_bytes"] + 2)                 # + EOF + EPSILON, zero-padded (framework.rs:997-1008)
    doc_log = K.logmn(udoc_len)
    safa = K.SafaShape(max_offset=udoc_len, **cfg["safa"])
    table = K.next_power_of_two(safa.num_edges)
    hybrid_len = 2 * K.next_power_of_two(max(table, udoc_len)) if cfg["hybrid"] else None   # r1cs.rs:481-487
    batch = cfg["batch"] or max(2, K.opt_cost_model_select(safa, udoc_len, cfg["hybrid"], hybrid_len, False, cfg["trace"]))   # r1cs.rs:489-513 (> 1)
    if cfg["merkle"]:
        # costs.rs has no Merkle term: the document lookups of nl_doc are replaced by b Merkle paths of log2 N Poseidon
        # hashes each (nova.rs:392-547).  Extrapolated with the model's own sponge-block constant (288, costs.rs:132).
        step = (K.nl(batch, table, False) + K.lookup_idxs(safa.num_states, batch) + K.cursor_circuit(udoc_len, batch, safa.max_offset)
                + K.stack_circuit(safa.num_states, udoc_len, safa.max_branches, safa.max_stack) + batch * doc_log * 288)
    else:
        step = K.full_round_cost_model(safa, batch, udoc_len, cfg["hybrid"], hybrid_len, False)
    steps = K.n_foldings(cfg["trace"], batch)
    primary = K.V1 + step
    table_log = K.logmn(hybrid_len) if cfg["hybrid"] else doc_log
    out = dict(name=cfg["name"], w1=primary, c1=primary, w2=K.V2, c2=K.V2, steps=steps, batch=batch, step_circuit_constraints=step,
               hyrax_row=0 if cfg["merkle"] else 1 << (doc_log - doc_log // 2), doc_log=0 if cfg["merkle"] else doc_log,
               symbol_bits=cfg["alphabet_bits"], table_log=0 if cfg["merkle"] else table_log, lookups=2 * batch if cfg["hybrid"] else batch,
               merkle_log=doc_log if cfg["merkle"] else 0,
               folded_cost=K.get_folded_cost(step, cfg["trace"], batch))
    return out


def main() -> None:
    shapes = [evaluate(c) for c in CONFIGS]
    doc = {
        "generated_by": "oracle/gen_replay_shapes.py (oracle/costs_oracle.py restates src/backend/costs.rs of eniac/Reef)",
        "constants": {"V1": K.V1, "V2": K.V2},
        "inputs": [{k: v for k, v in c.items()} for c in CONFIGS],
        "note": "MSM lengths are PREDICTIONS of Reef's cost model for assumed SAFA shapes (inputs), not measurements of a Reef run",
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
