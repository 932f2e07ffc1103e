"""Restatement of Reef's analytic constraint-count model.  TEST INFRASTRUCTURE ONLY.

Follows eniac/Reef src/backend/costs.rs function by function (citations below are lines of that file).  The
model decides the length of every per-step MSM of `reef --prove`: the primary step circuit has about
V1 + full_round_cost_model(...) constraints (its commitment key, |W1| and |C1| are of that length), the secondary
one about V2 (`get_folded_cost`, :168-179).  `oracle/gen_replay_shapes.py` evaluates it for BASELINE.json's
configs and writes tests/golden/replay_shapes.json, which the replay harness (reef_amd/csrc/host/reef_replay.cpp)
reads -- no MSM length is typed into the harness.

Pinned by the constants the reference holds (tests/test_costs_oracle.py): V1, V2 (:7-8), the Poseidon gadget
costs 578 / 288 / 290 (:120,132,136) and hand-worked values of each function.  The reference holds no expected
total for any document, so the totals are only as good as the SAFA shape fed in, which is an INPUT here
(`SafaShape`): Reef's frontend is not restated.

Rust semantics kept on purpose: `logmn` rounds an f32 logarithm up (:10-15), `n_sponge` floors an f32 quotient
(:127-131), integer divisions truncate.
"""
from __future__ import annotations

import math
import struct
from dataclasses import dataclass

V2 = 11376   # costs.rs:7  (secondary circuit: Nova's verifier circuit on the other curve)
V1 = 10347   # costs.rs:8  (primary circuit: the augmented-circuit overhead on top of the step circuit)


def _f32(x: float) -> float:
    return struct.unpack("f", struct.pack("f", x))[0]


def logmn(mn: int) -> int:
    """costs.rs:10-15: 1 for 1, else ceil(log2(mn as f32))."""
    if mn == 1:
        return 1
    return int(math.ceil(_f32(math.log2(_f32(float(mn))))))


def get_padding(solution_len: int, batch_size: int) -> int:
    """costs.rs:17-24."""
    modlen = solution_len + 1
    epsilon_to_add = batch_size - (modlen % batch_size)
    if modlen % batch_size == 0:
        epsilon_to_add = 0
    return epsilon_to_add + 1


def lookup_idxs(n_states: int, batch_size: int) -> int:
    """costs.rs:26-32."""
    bit_limit = logmn(n_states) + 1
    v_i = 5
    in_overflow = bit_limit * (2 * batch_size + 1)
    out_overflow = bit_limit * 3
    return in_overflow + out_overflow + v_i


def num_cqs(batch_size: int, log_mn: int) -> int:
    """costs.rs:59 / :116: ceil(batch_size * log_mn / 254) -- lookup indices packed 254 bits per field element."""
    return int(math.ceil((batch_size * log_mn) / 254.0))


def nl_nohash(batch_size: int, table_size: int) -> int:
    """costs.rs:34-64."""
    log_mn = logmn(table_size)
    cost = 0
    cost += batch_size + 1                         # multiplications (:39)
    cost += log_mn * 2                             # sum-check additions (:42)
    cost += (batch_size + 1) * (2 * log_mn)        # eq calc (:45)
    cost += (batch_size + 1) * (log_mn - 1)        # combine eqs (:48)
    cost += batch_size + 1                         # horners (:51)
    cost += 1                                      # mult by Tj (:54)
    cost += num_cqs(batch_size, log_mn)            # combine qs for Fiat-Shamir (:57-61)
    return cost


def nlookup_cost_hash(batch_size: int, table_size: int, hybrid: bool) -> int:
    """costs.rs:114-140: Poseidon gadget costs 578 + 288 per sponge block + 290 per sum-check round."""
    log_mn = logmn(table_size)
    cqs = num_cqs(batch_size, log_mn)
    cost = 578                                     # :120
    if log_mn + batch_size + cqs > 5:              # running claim (:123)
        num = _f32(float(log_mn + cqs + batch_size - 5))
        if hybrid:
            num = _f32(num + 1.0)
        n_sponge = int(math.floor(_f32(num / 4.0)))
        if n_sponge == 0:
            n_sponge += 1
        cost += n_sponge * 288                     # :132
    cost += log_mn * 290                           # sum-check Poseidon hashes (:136)
    return cost


def nl(batch_size: int, table_size: int, hybrid: bool) -> int:
    """costs.rs:66-70."""
    return nlookup_cost_hash(batch_size, table_size, hybrid) + nl_nohash(batch_size, table_size)


def q_ordering(table_size: int, batch_size: int, hybrid: bool, project: bool) -> int:
    """costs.rs:72-82."""
    total = logmn(table_size)
    if hybrid:
        total += 1
    if project:
        total += 1
    return total * batch_size


def nl_doc(batch_size: int, table_size: int, hybrid: bool, project: bool) -> int:
    """costs.rs:84-88."""
    return q_ordering(table_size, batch_size, hybrid, project) + nl(batch_size, table_size, hybrid)


def cursor_circuit(doc_len: int, batch_size: int, max_offset: int) -> int:
    """costs.rs:90-99."""
    cursor_plus = 1
    bitlimit = logmn(max(doc_len, max_offset)) + 1
    ite = 3 + 3 * bitlimit
    cur_overflow = bitlimit * (2 * batch_size + 1)
    min_offset_leq = bitlimit * 3 * batch_size
    max_offset_geq = bitlimit * 2 * batch_size
    upper_overflow = bitlimit * (batch_size + 1)
    return cursor_plus + cur_overflow + min_offset_leq + max_offset_geq + upper_overflow + ite


def stack_circuit(n_states: int, doc_len: int, max_branches: int, max_stack: int) -> int:
    """costs.rs:101-112."""
    log_states = logmn(n_states)
    bitlimit = logmn(doc_len) + 1
    push = 7 + max_branches * (3 + 2 * log_states + max_stack * 14) + log_states
    pop = 4 + max_stack * 7 + 4 * bitlimit
    ite = 27
    stack_ptr = 3
    not_forall = 14
    return push + pop + ite + stack_ptr + not_forall


@dataclass(frozen=True)
class SafaShape:
    """What the model reads off Reef's SAFA and solver (frontend/safa.rs), given here as inputs."""
    num_states: int
    num_edges: int
    max_offset: int
    max_branches: int
    max_stack: int


def next_power_of_two(x: int) -> int:
    return 1 if x <= 1 else 1 << (x - 1).bit_length()


def merkle_gadget(batch_size: int, doc_len: int) -> int:
    """NOT in costs.rs (it has no --merkle term): the constraints of NFAStepCircuit::eval_merkle, counted by hand from the gadget itself
    (src/backend/nova.rs:392-511, merkle_circuit :513-547, select :1404-1429) for `batch_size` document lookups in a tree over `doc_len`
    symbols (src/backend/merkle_tree.rs:25-78: level 0 hashes PAIRS of (index, symbol), so a path has 1 leaf hash + ceil(log2 doc_len) - 1
    inner hashes).  Per lookup:
      leaf      4 selects (one R1CS row each: :1421-1426) + one sponge of [Absorb(4), Squeeze(1)] + ensure_allocated (1 row, :538-542)
      per level 2 selects + one sponge of [Absorb(2), Squeeze(1)] + ensure_allocated
      root      1 equality row (:503-508)
    A sponge is ONE width-5 permutation either way (arity U4: 8 full rounds x 5 S-boxes + 56 partial rounds, x^5 = 3 rows each) = 288 rows --
    the constant costs.rs itself uses per sponge block (:132).  Allocations (root, the fillers, the l/r bits) add variables, not rows."""
    sponge = 288 + 1
    levels = max(1, logmn(next_power_of_two(doc_len)))            # hashes on a path
    per_lookup = (4 + sponge) + (levels - 1) * (2 + sponge) + 1
    return batch_size * per_lookup


def full_round_cost_model(safa: SafaShape, batch_size: int, doc_len: int, hybrid: bool, hybrid_len: int | None, project: bool) -> int:
    """costs.rs:142-166."""
    dlen_pow2 = next_power_of_two(doc_len)
    safa_pow2 = next_power_of_two(safa.num_edges)
    lookup_cost = lookup_idxs(safa.num_states, batch_size)
    if hybrid:
        assert hybrid_len is not None
        total_nl_cost = nl_doc(batch_size * 2, hybrid_len, hybrid, project) + lookup_cost
    else:
        nl_cost = nl(batch_size, safa_pow2, False)
        commit_cost = nl_doc(batch_size, dlen_pow2, hybrid, project)
        total_nl_cost = nl_cost + lookup_cost + commit_cost
    cursor_cost = cursor_circuit(dlen_pow2, batch_size, safa.max_offset)
    stack_cost = stack_circuit(safa.num_states, dlen_pow2, safa.max_branches, safa.max_stack)
    return total_nl_cost + stack_cost + cursor_cost


def get_folded_cost(cost: int, solution_lens: list[int], batch_size: int) -> int:
    """costs.rs:168-179: 2 * n_fold * (V1 + V2 + c) + 8 * (V1 + c)."""
    n_folding = sum(int(math.ceil(_f32(_f32(float(x)) / _f32(float(batch_size))))) for x in solution_lens)
    return 2 * n_folding * (V1 + V2 + cost) + 8 * (V1 + cost)


def n_foldings(solution_lens: list[int], batch_size: int) -> int:
    return sum(int(math.ceil(_f32(_f32(float(x)) / _f32(float(batch_size))))) for x in solution_lens)


def opt_cost_model_select(safa: SafaShape, doc_len: int, hybrid: bool, hybrid_len: int | None, project: bool, solution: list[int]) -> int:
    """costs.rs:207-244: the batch size with the least folded cost among 1 .. sum(solution)."""
    best, best_cost = 0, None
    for n in range(1, sum(solution) + 1):
        c = get_folded_cost(full_round_cost_model(safa, n, doc_len, hybrid, hybrid_len, project), solution, n)
        if best_cost is None or c < best_cost:
            best, best_cost = n, c
    return best
