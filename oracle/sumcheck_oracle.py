"""Big-integer restatement of Reef's host sum-check helpers (row N2 of SURVEY.md 8f).
TEST INFRASTRUCTURE ONLY -- only tests/, smoke() and bench.py's cpu_baseline leg may import it.

Each function follows the reference function of the same name in
/root/reference/src/backend/r1cs_helper.rs line by line (plain Python ints instead of
rug::Integer; `% Q` instead of rem_floor(modulus).keep_bits(255)):

    linear_mle_product        r1cs_helper.rs:441-506   (split here into the coefficient half,
                                                        :455-476, and the table fold, :491-503; the
                                                        Poseidon challenge :478-489 stays with the
                                                        caller, as it does on the host in Reef)
    gen_eq_table              r1cs_helper.rs:508-544
    prover_mle_partial_eval   r1cs_helper.rs:551-634
    verifier_mle_eval         r1cs_helper.rs:637-641

PARITY PINNED BY THE REFERENCE'S OWN TESTS: tests/test_sumcheck_oracle.py replays the known-answer
and identity checks of `mle_linear_basic` (r1cs.rs:2411-2515) and `mle_partial` (r1cs.rs:2517-2578)
on the same inputs.  The only substitution: the sponge challenges (neptune Poseidon, not available
here) are replaced by fixed field elements -- every assertion of those tests holds for any
challenge, and is checked for several.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

# cfg().field().modulus(): the Pallas scalar field, r1cs_helper.rs:37-38
Q = 28948022309329048855892746252171976963363056481941647379679742748393362948097


def linear_mle_coeffs(table_t: Sequence[int], table_eq: Sequence[int], ell: int, i: int, q: int = Q) -> Tuple[int, int, int]:
    """First half of linear_mle_product (r1cs_helper.rs:448-476): returns (xsq, x, con)."""
    pow_ = 2 ** (ell - i)
    assert len(table_t) == 2 ** ell and len(table_eq) == 2 ** ell
    xsq = x = con = 0
    for b in range(pow_):
        ti_0, ti_1 = table_t[b], table_t[b + pow_]
        ei_0, ei_1 = table_eq[b], table_eq[b + pow_]
        t_slope = ti_1 - ti_0
        e_slope = ei_1 - ei_0
        xsq += t_slope * e_slope
        x += e_slope * ti_0
        x += t_slope * ei_0
        con += ti_0 * ei_0
    return xsq % q, x % q, con % q


def linear_mle_fold(table_t: List[int], table_eq: List[int], ell: int, i: int, r_i: int, q: int = Q) -> None:
    """Second half of linear_mle_product (r1cs_helper.rs:491-503), in place."""
    pow_ = 2 ** (ell - i)
    for b in range(pow_):
        table_t[b] = (table_t[b] * (1 - r_i) + table_t[b + pow_] * r_i) % q
        table_eq[b] = (table_eq[b] * (1 - r_i) + table_eq[b + pow_] * r_i) % q


def gen_eq_table(rs: Sequence[int], qs: Sequence[int], last_q: Sequence[int], q: int = Q) -> List[int]:
    """r1cs_helper.rs:508-544."""
    ell = len(last_q)
    t_len = 2 ** ell
    assert len(rs) == len(qs) + 1
    eq_t = [0] * t_len
    for i in range(len(qs)):
        eq_t[qs[i]] += rs[i]
    for i in range(t_len):
        term = rs[len(qs)]
        for j in reversed(range(ell)):
            xi = (i >> j) & 1
            term *= xi * last_q[j] + (1 - xi) * (1 - last_q[j])
        eq_t[i] = (eq_t[i] + term) % q
    return eq_t


def prover_mle_partial_eval(prods: Sequence[int], x: Sequence[int], es: Sequence[int], for_t: bool,
                            last_q: Optional[Sequence[int]], q: int = Q) -> Tuple[int, int]:
    """r1cs_helper.rs:551-634.  x entries equal to -1 mark the "hole"."""
    m = len(x)
    if for_t:
        assert 2 ** (m - 1) <= len(prods) <= 2 ** m
        assert len(es) == len(prods)
    elif last_q is not None:
        assert len(es) + 1 == len(prods)
    hole_coeff = 0
    minus_coeff = 0
    for i in range(len(es) + 1):
        if i < len(es):
            prod = prods[i]
            next_hole_coeff = 0
            for j in reversed(range(m)):
                ej = (es[i] >> j) & 1
                if x[m - j - 1] == -1:
                    next_hole_coeff = ej
                else:
                    prod *= x[m - j - 1] if ej == 1 else (1 - x[m - j - 1])
            if next_hole_coeff == 1:
                hole_coeff += prod
            else:
                minus_coeff += prod
        elif last_q is not None:
            prod = prods[i]
            next_hole_coeff = 1
            next_minus_coeff = 1
            for j in range(m):
                ej = last_q[j]
                if x[j] == -1:
                    next_hole_coeff = ej
                    next_minus_coeff = 1 - ej
                else:
                    prod *= ej * x[j] + (1 - ej) * (1 - x[j])
            hole_coeff += prod * next_hole_coeff
            minus_coeff += prod * next_minus_coeff
    hole_coeff -= minus_coeff
    return hole_coeff % q, minus_coeff % q


def verifier_mle_eval(table: Sequence[int], qpt: Sequence[int], q: int = Q) -> int:
    """r1cs_helper.rs:637-641."""
    return prover_mle_partial_eval(table, qpt, list(range(len(table))), True, None, q)[1]


# ---- the Fiat-Shamir sponge of a folding step (round 4) -------------------------------------------------------------------
class Sponge:
    """neptune's Sponge<F, U4> in Mode::Simplex as Reef drives it through the SpongeAPI [R: the crate is not in the reference tree]:
    state = [tag, 0, 0, 0, 0] with the tag derived from the IO pattern (src/backend/r1cs.rs:2260-2284: Absorb(k), Squeeze(1), then
    (Absorb(3), Squeeze(1)) per sum-check round); absorb adds elements into the rate positions 1..4, permuting when they are full;
    squeeze permutes first if anything was absorbed since the last permutation and reads the rate positions in order.  The permutation
    is oracle/merkle_oracle.py's (the published Poseidon round structure); constants and tag are the caller's.  PARITY UNPINNED until
    tests/golden/rust_pin.json exists (tools/rust_pin): tests/test_pin_from_rust.py then replays neptune's own transcript."""

    def __init__(self, params, tag: int):
        from .merkle_oracle import poseidon_permute
        self._permute = lambda st: poseidon_permute(st, params)
        self.p = params
        self.rate = params.t - 1
        self.state = [tag % params.m] + [0] * self.rate
        self.absorb_pos = 0
        self.squeeze_pos = 0
        self.dirty = False            # absorbed since the last permutation

    def absorb(self, elems: Sequence[int]) -> None:
        for e in elems:
            if self.absorb_pos == self.rate:
                self.state = self._permute(self.state)
                self.absorb_pos = 0
            self.state[1 + self.absorb_pos] = (self.state[1 + self.absorb_pos] + e) % self.p.m
            self.absorb_pos += 1
            self.dirty = True
        self.squeeze_pos = self.rate  # the next squeeze starts from a fresh permutation

    def squeeze(self, n: int = 1) -> List[int]:
        out = []
        for _ in range(n):
            if self.dirty or self.squeeze_pos == self.rate:
                self.state = self._permute(self.state)
                self.absorb_pos = 0
                self.squeeze_pos = 0
                self.dirty = False
            out.append(self.state[1 + self.squeeze_pos])
            self.squeeze_pos += 1
        return out


def linear_mle_product(table_t: List[int], table_eq: List[int], ell: int, i: int, sponge: Sponge, q: int = Q) -> Tuple[int, int, int, int]:
    """The whole of linear_mle_product (r1cs_helper.rs:441-506): sums, absorb (con, x, xsq) in that order (:478-482), squeeze the
    challenge (:485-488), fold both tables.  Returns (r_i, xsq, x, con) like the reference."""
    xsq, x, con = linear_mle_coeffs(table_t, table_eq, ell, i, q)
    sponge.absorb([con, x, xsq])
    r_i = sponge.squeeze(1)[0]
    linear_mle_fold(table_t, table_eq, ell, i, r_i, q)
    return r_i, xsq, x, con
