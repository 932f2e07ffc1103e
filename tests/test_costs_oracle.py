"""oracle/costs_oracle.py against the constants and formulas of the reference's src/backend/costs.rs, and the replay shapes
derived from it (tests/golden/replay_shapes.json) against a fresh evaluation.

The reference holds no expected total for any document (no test calls the model with a pinned answer), so what is pinned
is: the constants it hard-codes (costs.rs:7-8,120,132,136), hand-worked values of every function, and that the committed
shapes file is what oracle/gen_replay_shapes.py produces today."""
import json
import os

from oracle import costs_oracle as K
from oracle import gen_replay_shapes as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_constants_of_costs_rs():
    assert (K.V1, K.V2) == (10347, 11376)                     # costs.rs:7-8
    # a table of one entry, batch 1: log_mn = 1, one packed q, no running-claim sponge (1 + 1 + 1 <= 5)
    assert K.nlookup_cost_hash(1, 1, False) == 578 + 1 * 290   # costs.rs:120,136
    # log_mn = 4, batch 4: 4 + 4 + 1 - 5 = 4 -> one sponge block of 288 (costs.rs:123-132)
    assert K.nlookup_cost_hash(4, 16, False) == 578 + 288 + 4 * 290
    assert K.nlookup_cost_hash(4, 16, True) == 578 + 288 + 4 * 290        # (4 + 1) / 4 floors to 1
    assert K.nlookup_cost_hash(8, 16, False) == 578 + 2 * 288 + 4 * 290   # 4 + 1 + 8 - 5 = 8 -> two blocks


def test_logmn_and_padding():
    assert [K.logmn(x) for x in (1, 2, 3, 4, 5, 1 << 21, (1 << 21) + 1)] == [1, 1, 2, 2, 3, 21, 21]   # f32: 2^21 + 1 rounds to 2^21 (costs.rs:10-15)
    assert K.logmn((1 << 21) + 3) == 22
    assert K.get_padding(10, 4) == 2 and K.get_padding(11, 4) == 1 and K.get_padding(3, 4) == 1   # costs.rs:17-24


def test_hand_worked_pieces():
    assert K.lookup_idxs(12, 4) == 5 * 9 + 5 * 3 + 5                     # bit_limit = 4 + 1 (costs.rs:26-32)
    # nl_nohash(batch 4, table 16): 5 + 8 + 5*8 + 5*3 + 5 + 1 + ceil(16/254) = 75
    assert K.nl_nohash(4, 16) == 75
    assert K.q_ordering(1 << 21, 4, False, False) == 84 and K.q_ordering(1 << 26, 64, True, True) == 28 * 64
    # cursor_circuit(2^21, 4, 2^21): bitlimit 22: 1 + 22*9 + 22*12 + 22*8 + 22*5 + 69
    assert K.cursor_circuit(1 << 21, 4, 1 << 21) == 1 + 198 + 264 + 176 + 110 + 69
    # stack_circuit(12, 2^21, 1, 1): log_states 4, bitlimit 22
    assert K.stack_circuit(12, 1 << 21, 1, 1) == (7 + (3 + 8 + 14) + 4) + (4 + 7 + 88) + 27 + 3 + 14
    assert K.get_folded_cost(1000, [11], 4) == 2 * 3 * (K.V1 + K.V2 + 1000) + 8 * (K.V1 + 1000)   # costs.rs:168-179


def test_full_round_model_is_the_sum_of_its_parts():
    safa = K.SafaShape(num_states=12, num_edges=12, max_offset=1 << 21, max_branches=1, max_stack=1)
    b, n = 4, 1 << 21
    want = (K.nl(b, 16, False) + K.lookup_idxs(12, b) + K.nl_doc(b, n, False, False) + K.cursor_circuit(n, b, 1 << 21)
            + K.stack_circuit(12, n, 1, 1))
    assert K.full_round_cost_model(safa, b, n, False, None, False) == want
    hy = K.full_round_cost_model(safa, b, n, True, 1 << 22, False)
    assert hy == K.nl_doc(2 * b, 1 << 22, True, False) + K.lookup_idxs(12, b) + K.cursor_circuit(n, b, 1 << 21) + K.stack_circuit(12, n, 1, 1)
    best = K.opt_cost_model_select(safa, n, False, None, False, [11])
    costs = {bb: K.get_folded_cost(K.full_round_cost_model(safa, bb, n, False, None, False), [11], bb) for bb in range(1, 12)}
    assert costs[best] == min(costs.values())


def test_committed_shapes_are_what_the_model_gives():
    with open(os.path.join(ROOT, "tests", "golden", "replay_shapes.json")) as f:
        doc = json.load(f)
    assert doc["constants"] == {"V1": K.V1, "V2": K.V2}
    assert doc["shapes"] == [G.evaluate(c) for c in G.CONFIGS]
    for s in doc["shapes"]:
        assert s["w1"] == K.V1 + s["step_circuit_constraints"] and s["w2"] == K.V2 and s["steps"] >= 1 and s["batch"] > 1   # r1cs.rs:513
    # every automaton is DERIVED (oracle/safa_shape.py, tests/test_safa_shape.py) from a regex with a source; nothing about it is typed in
    by_name = {s["name"]: s for s in doc["shapes"]}
    assert all("safa" not in c and c["source"] and c["regex"] for c in doc["inputs"])
    cfg4 = by_name["cfg4_16MiB_dna_hybrid_b32"]
    assert (cfg4["safa_states"], cfg4["safa_edges"], cfg4["solution_lens"], cfg4["steps"]) == (64, 309, [63], 2)     # ceil(63 / 32) folding steps
    assert by_name["cfg4b_16MiB_dna_three_literals_hybrid_b32"]["steps"] == 10


def test_merkle_gadget_hand_count():
    """cfg5's Merkle term (costs.rs has none): the hand count of NFAStepCircuit::eval_merkle (nova.rs:392-511) -- per lookup a leaf hash behind four
    selects, an inner hash behind two selects per level, one root equality; a hash = one width-5 Poseidon permutation (288 rows, the constant of
    costs.rs:132) + ensure_allocated."""
    from oracle import costs_oracle as K
    assert K.merkle_gadget(1, 2) == 4 + 289 + 1                              # two symbols: the leaf hash is the root
    assert K.merkle_gadget(1, 4) == (4 + 289) + (2 + 289) + 1
    assert K.merkle_gadget(1, 1 << 27) == (4 + 289) + 26 * (2 + 289) + 1 == 7860
    assert K.merkle_gadget(122, (1 << 26) + 2) == 122 * 7860                 # cfg5: 64 MiB + EOF + EPSILON pads to 2^27 symbols
    assert K.merkle_gadget(3, 5) == 3 * K.merkle_gadget(1, 8)                # ragged documents pay for the padded height
    import json
    import os
    doc = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "replay_shapes.json")))
    cfg5 = next(s for s in doc["shapes"] if s["name"].startswith("cfg5"))
    assert cfg5["merkle_gadget_constraints"] == 122 * 7860 and "nova.rs:392-511" in cfg5["merkle_gadget_basis"]
