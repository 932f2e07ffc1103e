"""CPU restatement of row N3 (bound rows and evaluation of a multilinear table).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

What it restates
  * NLDocCommitment::proof_dot_prod_prover, src/backend/commitment.rs:287-405:
      v' = doc_poly.evaluate(&running_q)                         (:357)
      hyrax_gen.prove_eval(&doc_poly, .., &running_q, ..)        (:371-379 / :383-391)
    whose first step binds the left half of the point to the rows of the matrix view Hyrax
    committed to (compute_factored_lens, :173-174): LZ = L^T Z, then proves <LZ, R> = v'.
  * verifier_mle_eval(table, q'), commitment.rs:236 -> r1cs_helper.rs:637-641.

The polynomial code itself (MultilinearPolynomial / EqPolynomial / HyraxPC) lives in the nova-snark
fork (github.com/sga001/Nova, default branch, no rev) which is not under /root/reference, so the
definition used here is the mathematical one -- sum_i eq(point, i) * Z[i] -- with the variable
order taken from Reef's own prover_mle_partial_eval (r1cs_helper.rs:577-592: x[0] pairs with the
most significant index bit).  Reef's consistency proof only verifies if both agree
(r1cs.rs:2701-2723 round trip), and the reference's `mle_partial` known-answer test
(r1cs.rs:2517-2578) pins the order: tests/test_mle_oracle.py replays it against this file and
checks this file against the line-by-line restatement in sumcheck_oracle.verifier_mle_eval.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

from .pasta_oracle import P, Q  # noqa: F401  (moduli: Q = scalar field of Pallas, P = of Vesta)


def eq_evals(point: Sequence[int], mod: int) -> List[int]:
    """eq[t] = prod_k (bit_k(t) ? r_k : 1 - r_k), bit_k = bit (len-1-k) of t (r_0 = most significant)."""
    out = [1]
    for r in point:
        r %= mod
        nxt = []
        for v in out:
            nxt.append(v * (1 - r) % mod)
            nxt.append(v * r % mod)
        out = nxt
    return out


def bound_rows(z: Sequence[int], point: Sequence[int], left_vars: int, mod: int) -> Tuple[List[int], int]:
    """(LZ, eval): LZ[j] = sum_i L[i] * Z[i*cols + j], eval = <LZ, R>; z is zero-padded to 2^len(point)."""
    m = len(point)
    cols = 1 << (m - left_vars)
    assert len(z) <= 1 << m
    L = eq_evals(point[:left_vars], mod)
    R = eq_evals(point[left_vars:], mod)
    lz = [0] * cols
    for idx, v in enumerate(z):
        if v:
            i, j = divmod(idx, cols)
            lz[j] = (lz[j] + L[i] * v) % mod
    ev = sum(a * b for a, b in zip(lz, R)) % mod
    return lz, ev


def evaluate(z: Sequence[int], point: Sequence[int], mod: int) -> int:
    """The multilinear extension of z (zero-padded) at point: what doc_poly.evaluate returns."""
    return bound_rows(z, point, len(point) // 2, mod)[1]
