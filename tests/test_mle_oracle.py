"""Row N3 oracle vs the reference's own known answers (CPU only)."""
import itertools

from oracle import mle_oracle, sumcheck_oracle
from oracle.pasta_oracle import P, Q, SplitMix64


def test_reference_mle_partial_kat_boolean_points():
    """r1cs.rs:2517-2578 (`mle_partial`): at boolean x the evaluation is table[x1*4 + x2*2 + x3]."""
    table = [1, 3, 8, 2, 9, 5, 13, 4]
    for x in itertools.product((0, 1), repeat=3):
        e = x[0] * 4 + x[1] * 2 + x[2]
        assert mle_oracle.evaluate(table, x, Q) == table[e]
        for left in range(4):
            lz, ev = mle_oracle.bound_rows(table, x, left, Q)
            assert ev == table[e]
            cols = 1 << (3 - left)
            row = sum(b << (left - 1 - k) for k, b in enumerate(x[:left]))
            assert lz == table[row * cols:(row + 1) * cols]     # a boolean left half selects a row


def test_matches_line_by_line_restatement_of_verifier_mle_eval():
    """verifier_mle_eval (r1cs_helper.rs:637-641) is restated line by line in sumcheck_oracle."""
    rng = SplitMix64(77)
    for mod in (Q, P):
        for m in (1, 2, 5, 7):
            table = [rng.next() % 300 if m < 7 else (rng.next() << 64 | rng.next()) % mod for _ in range(1 << m)]
            point = [(rng.next() << 192 | rng.next() << 128 | rng.next() << 64 | rng.next()) % mod for _ in range(m)]
            want = sumcheck_oracle.verifier_mle_eval(table, point, mod)
            for left in (0, m // 2, m):
                assert mle_oracle.bound_rows(table, point, left, mod)[1] == want
            short = table[:(1 << m) - 3] if m > 2 else table       # zero padding == explicit zeros
            padded = short + [0] * ((1 << m) - len(short))
            assert mle_oracle.evaluate(short, point, mod) == mle_oracle.evaluate(padded, point, mod)
