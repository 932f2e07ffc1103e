"""Pins oracle/sumcheck_oracle.py to the reference's own tests of the host sum-check helpers:
`mle_linear_basic` (src/backend/r1cs.rs:2411-2515) and `mle_partial` (r1cs.rs:2517-2578), same
inputs, same assertions.  The neptune Poseidon challenges are replaced by fixed field elements
(every assertion below is challenge-independent; several challenge sets are tried)."""
import itertools

import pytest

from oracle.sumcheck_oracle import (Q, gen_eq_table, linear_mle_coeffs, linear_mle_fold, prover_mle_partial_eval,
                                    verifier_mle_eval)

CHALLENGES = [
    [5, 7, 11],
    [Q - 1, 2, Q // 3],
    [0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % Q, 1, 0],
]


@pytest.mark.parametrize("sc_challenges", CHALLENGES)
def test_mle_linear_basic(sc_challenges):
    evals = [2, 3, 5, 7, 9, 13, 17, 19]                       # r1cs.rs:2415-2424
    table = list(evals)
    qs = [2, 1, 7]                                            # r1cs.rs:2428
    last_q = [2, 3, 5]                                        # r1cs.rs:2430
    claims = [3, 9, 27, 81]                                   # r1cs.rs:2433-2438
    term = sum(evals[qs[i]] * claims[i] for i in range(len(qs)))
    eq_a = gen_eq_table(claims, qs, list(reversed(last_q)))   # r1cs.rs:2445
    _, running_v = prover_mle_partial_eval(evals, last_q, list(range(len(evals))), True, None)
    term += running_v * claims[3]
    claim = sum(t * e for t, e in zip(evals, eq_a))
    assert term % Q == claim % Q                              # r1cs.rs:2460-2466
    sc_rs = []
    for i in range(1, 4):
        xsq, x, con = linear_mle_coeffs(evals, eq_a, 3, i)
        assert claim % Q == (2 * con + x + xsq) % Q           # g(0) + g(1), r1cs.rs:2489-2493
        r_i = sc_challenges[i - 1]
        linear_mle_fold(evals, eq_a, 3, i, r_i)
        claim = (xsq * r_i * r_i + x * r_i + con) % Q
        sc_rs.append(r_i)
    _, next_running_v = prover_mle_partial_eval(table, sc_rs, list(range(len(table))), True, None)
    _, eq_term = prover_mle_partial_eval(claims, sc_rs, qs, False, last_q)
    assert claim == (eq_term * next_running_v) % Q            # r1cs.rs:2507-2513
    # what the GPU path exploits: after all rounds the folded table IS the running claim
    assert evals[0] == next_running_v


def test_mle_partial():
    table = [1, 3, 8, 2, 9, 5, 13, 4]                         # r1cs.rs:2521-2530
    for x_1, x_2, x_3 in itertools.product([0, 1, -1], repeat=3):
        coeff, con = prover_mle_partial_eval(table, [x_1, x_2, x_3], list(range(len(table))), True, None)
        holes = (x_1 == -1) + (x_2 == -1) + (x_3 == -1)
        if ((x_1 == -1) ^ (x_2 == -1) ^ (x_3 == -1)) and not (x_1 + x_2 + x_3 == -3):
            if x_1 == -1:
                assert (coeff + con) % Q == table[4 + x_2 * 2 + x_3]
                assert con == table[x_2 * 2 + x_3]
            elif x_2 == -1:
                assert (coeff + con) % Q == table[x_1 * 4 + 2 + x_3]
                assert con == table[x_1 * 4 + x_3]
            elif x_3 == -1:
                assert (coeff + con) % Q == table[x_1 * 4 + x_2 * 2 + 1]
                assert con == table[x_1 * 4 + x_2 * 2]
        elif holes == 0:
            assert table[x_1 * 4 + x_2 * 2 + x_3] == con
    assert verifier_mle_eval(table, [1, 0, 1]) == table[5]


def test_c_port_matches_python_oracle(cref):
    """The C restatement of a sum-check round (the timed CPU baseline of row N2) == the oracle."""
    import numpy as np
    from oracle.pasta_oracle import SplitMix64, uniform_scalar
    ell = 8
    rng = SplitMix64(5)
    t = [uniform_scalar(rng, Q) for _ in range(1 << ell)]
    e = [uniform_scalar(rng, Q) for _ in range(1 << ell)]
    T = np.array([[(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)] for v in t], dtype=np.uint64)
    E = np.array([[(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)] for v in e], dtype=np.uint64)
    cref.sc_to_mont(1, T)
    cref.sc_to_mont(1, E)
    for i in range(1, ell + 1):
        r = uniform_scalar(rng, Q)
        exp = linear_mle_coeffs(t, e, ell, i)
        assert cref.sc_round(1, T, E, 1 << (ell - i), r) == exp
        linear_mle_fold(t, e, ell, i, r)
    cref.sc_from_mont(1, T)
    assert cref.limbs_to_int(T[0]) == t[0]
