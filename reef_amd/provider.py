"""Host-side mirror of the nova-snark provider interface that Reef calls for its commitments.

Reef itself defines no plugin/FFI interface for this path: it calls the Rust
generics of nova-snark (git dep sga001/Nova, not vendored).  This module mirrors the *names and
argument meaning* of the pieces Reef reaches, on top of the C ABI, so that tests read like the
reference's call sites and so that a maintainer can see what each Rust method maps to:

    reference call site (eniac/Reef)                         here
    -------------------------------------------------------  --------------------------------
    G::vartime_multiscalar_mul(scalars, bases)        [R]    vartime_multiscalar_mul(...)
    CommitmentGens::<G>::new(label, n) / new_with_..  [R]    CommitmentGens.new(curve, label, n, <hash-to-curve constants>) /
        (framework.rs:297-303, commitment.rs:176-180)         .new_with_blinding_gen(.., h): derived on the GPU (row N1);
                                                               CommitmentGens(curve, bases [, h]) for bases that are given
    MerkleCommitment::new(&doc, &pc), path_wits, make_wits   reef_amd.merkle.MerkleCommitment (row N4: tree on the GPU, openings
        (merkle_tree.rs:25-190)                               looked up on the host)
    CE::commit(&gens, &v, &blind)                      [R]    CommitmentGens.commit(v, blind)
        (commitment.rs:350,361,422,430)
    gens.fold(w1, w2) / split_at / combine             [R]    CommitmentGens.fold / split_at / combine  (the points), or
        (ipa_pc inside CompressedSNARK::prove, framework.rs:695)      CommitmentGens.fold_lazy -> FoldedGens (fold recorded, never performed)
    HyraxPC::commit(&poly)                             [R]    HyraxPC.commit(poly)
        (commitment.rs:187)
    the same on the GPUs of a node, from one process          CommitmentGens(..., devices=[0, 1, ..]): device groups
    Commitment::compress()                             [R]    Commitment.compress()
        (commitment.rs:195,351,365,425,427,431)

[R] = recalled interface of an un-vendored crate; errors follow the reference's convention
(`assert!`/`panic!` there, exceptions here).  All group arithmetic runs in libreef_msm.so.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np

from . import msm

MODULI = {
    # scalar field of each curve (Pallas scalars live in Fq = the modulus Reef hard-codes at
    # src/backend/r1cs_helper.rs:37-38)
    msm.PALLAS: 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001,
    msm.VESTA: 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001,
}


def scalars_to_array(values: Sequence[int], curve) -> np.ndarray:
    """Canonical Python ints -> (n, 4) uint64 canonical limbs (pass is_mont=False downstream)."""
    q = MODULI[msm.curve_id(curve)]
    out = np.zeros((len(values), 4), dtype=np.uint64)
    for i, v in enumerate(values):
        v %= q
        for j in range(4):
            out[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


class Commitment:
    """A group element as the engine returns it (Jacobian, ABI layout)."""

    def __init__(self, curve, jac: np.ndarray):
        self.curve = msm.curve_id(curve)
        self.jac = np.ascontiguousarray(jac, dtype=np.uint64).reshape(12)

    def compress(self) -> bytes:
        return msm.compress(self.curve, self.jac)

    def to_affine(self) -> np.ndarray:
        return msm.normalize(self.curve, self.jac)[0][0]

    def __eq__(self, other) -> bool:  # equality on the canonical encoding, like the Rust side
        return isinstance(other, Commitment) and self.curve == other.curve and self.compress() == other.compress()

    def __add__(self, other: "Commitment") -> "Commitment":
        assert self.curve == other.curve
        return Commitment(self.curve, msm.sum_points(self.curve, np.stack([self.jac, other.jac])))


def vartime_multiscalar_mul(curve, scalars: np.ndarray, bases: np.ndarray, is_mont: bool = True) -> Commitment:
    """Group::vartime_multiscalar_mul: stateless, through the pasta-msm drop-in symbol."""
    if len(scalars) != len(bases):
        raise ValueError("scalars and bases differ in length")  # the Rust wrapper panics
    return Commitment(curve, msm.mult_pippenger(curve, bases, scalars, is_mont=is_mont))


class CommitmentGens:
    """nova-snark `CommitmentGens<G>`: a vector of generators (+ optional blinding generator h)
    resident on the GPU."""

    def __init__(self, curve, bases: np.ndarray, h: Optional[np.ndarray] = None, *, precompute: bool = True,
                 window_bits: int = 0, devices: Optional[Sequence[int]] = None, split: int = msm.SPLIT_WINDOWS):
        """devices: keep the key on several GPUs of this process (reef_msm_group_*, include/reef_msm.h section 5): commitments are split by
        Pippenger window (or by points) over them and Hyrax rows dealt out whole -- same points, whatever the devices."""
        self.curve = msm.curve_id(curve)
        self._devices = None if devices is None else [int(d) for d in devices]
        self._split = split
        self._group: Optional[msm.MsmGroup] = None
        self.bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
        self.h = None if h is None else np.ascontiguousarray(h, dtype=np.uint64).reshape(8)
        self._precompute = precompute
        self._window_bits = window_bits
        self._ctx: Optional[msm.MsmContext] = None

    @classmethod
    def new(cls, curve, label: bytes, n: int, *, a: int, b: int, z: int, iso: Sequence[int], dst: bytes, little_endian: bool = False,
            h: Optional[np.ndarray] = None, **kw) -> "CommitmentGens":
        """CommitmentGens::new(label, n) [R] (src/backend/framework.rs:297-303, commitment.rs:146-149, :176-180): the n generators are
        derived from the label on the GPU (row N1, reef_derive_generators: SHAKE256 stream on the host, the 2n maps to the curve and the
        batch normalisation on the device).  pasta_curves' hash-to-curve constants (a, b, z of the isogenous curve, the 13 isogeny
        coefficients, the domain separation string) are inputs: they live in an un-vendored crate."""
        from . import keygen
        return cls(curve, keygen.derive_generators(curve, label, n, a, b, z, iso, dst, little_endian), h, **kw)

    @classmethod
    def new_with_blinding_gen(cls, curve, label: bytes, n: int, blinding_gen: np.ndarray, **kw) -> "CommitmentGens":
        """The fork's CommitmentGens::new_with_blinding_gen(label, n, &h) [R] (commitment.rs:176-180): generators from the label, h given."""
        return cls.new(curve, label, n, h=blinding_gen, **kw)

    def __len__(self) -> int:
        return self.bases.shape[0]

    def _context(self) -> msm.MsmContext:
        if self._ctx is None:
            self._ctx = msm.MsmContext(self.curve, self.bases, window_bits=self._window_bits,
                                       bucket_groups=1 if self._precompute else 0)
        return self._ctx

    def _on_devices(self) -> Optional[msm.MsmGroup]:
        if self._devices is None:
            return None
        if self._group is None:
            self._group = msm.MsmGroup(self.curve, self.bases, self._devices, split=self._split, window_bits=self._window_bits,
                                       bucket_groups=1 if self._precompute else 0)
        return self._group

    def close(self) -> None:
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None
        if self._group is not None:
            self._group.close()
            self._group = None

    # CE::commit(&gens, &v, &blind)
    def commit(self, v: np.ndarray, blind: Optional[np.ndarray] = None, *, is_mont: bool = True) -> Commitment:
        v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
        if v.shape[0] > len(self):
            raise ValueError(f"vector of {v.shape[0]} scalars exceeds {len(self)} generators")
        grp = self._on_devices()
        if blind is None:
            return Commitment(self.curve, grp.msm(v, is_mont=is_mont) if grp is not None else self._context().msm(v, is_mont=is_mont))
        if self.h is None:
            raise ValueError("these generators have no blinding generator")
        b = np.ascontiguousarray(blind, dtype=np.uint64).reshape(1, 4)
        if grp is not None:
            if self._split != msm.SPLIT_WINDOWS:
                raise ValueError("a commitment with a blind needs the whole key on every device (split = windows)")
            return Commitment(self.curve, grp.msm_rows(v, 1, v.shape[0], is_mont=is_mont, blinds=b, h=self.h)[0])
        out = self._context().msm_rows(v, 1, v.shape[0], is_mont=is_mont, blinds=b, h=self.h)
        return Commitment(self.curve, out[0])

    # gens.fold(w1, w2): G'_i = w1*G_i + w2*G_{n/2+i}
    def fold(self, w1: int, w2: int) -> "CommitmentGens":
        n = len(self)
        if n % 2:
            raise ValueError("fold needs an even number of generators")
        folded = msm.fold(self.curve, self.bases, n // 2, w1, w2)
        return CommitmentGens(self.curve, folded, self.h, precompute=False)

    # the same without the scalar multiplications: the fold is recorded, commitments go over this (resident) key
    def fold_lazy(self, w1: int, w2: int) -> "FoldedGens":
        return FoldedGens(self).fold(w1, w2)

    def split_at(self, k: int) -> Tuple["CommitmentGens", "CommitmentGens"]:
        return (CommitmentGens(self.curve, self.bases[:k].copy(), self.h, precompute=False),
                CommitmentGens(self.curve, self.bases[k:].copy(), self.h, precompute=False))

    def combine(self, other: "CommitmentGens") -> "CommitmentGens":
        assert self.curve == other.curve
        return CommitmentGens(self.curve, np.concatenate([self.bases, other.bases]), self.h, precompute=False)


class FoldedGens:
    """`gens.fold(w1, w2)` that records the fold instead of performing it: the generators after k folds are fixed linear
    combinations of the resident key, so every commitment over them -- the cross terms of an IPA round over the halves of
    `split_at`, the last remaining generator -- is one MSM over the ORIGINAL pre-shifted key (reef_msm_folded).  Same
    methods as CommitmentGens where nova-snark's ipa_pc [R] uses them; `materialize()` performs the folds (K3) when the
    points themselves are wanted."""

    def __init__(self, root: "CommitmentGens", w1s=(), w2s=(), off: int = 0, length: Optional[int] = None):
        self.root, self.w1s, self.w2s, self.off = root, list(w1s), list(w2s), off
        n_k = len(root) >> len(self.w1s)
        if (n_k << len(self.w1s)) != len(root):
            raise ValueError("the key length must be a multiple of 2^folds")
        self.length = n_k - off if length is None else length
        if off < 0 or self.length < 0 or off + self.length > n_k:
            raise ValueError("slice outside the folded generators")
        self.curve, self.h = root.curve, root.h

    def __len__(self) -> int:
        return self.length

    def commit(self, v: np.ndarray, blind: Optional[np.ndarray] = None, *, is_mont: bool = True) -> Commitment:
        v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
        if v.shape[0] > self.length:
            raise ValueError(f"vector of {v.shape[0]} scalars exceeds {self.length} generators")
        c = Commitment(self.curve, self.root._context().msm_folded(v, self.w1s, self.w2s, self.off, is_mont=is_mont))
        if blind is None:
            return c
        if self.h is None:
            raise ValueError("these generators have no blinding generator")
        hg = CommitmentGens(self.curve, self.h.reshape(1, 8))
        try:
            return c + hg.commit(np.ascontiguousarray(blind, dtype=np.uint64).reshape(1, 4), is_mont=is_mont)
        finally:
            hg.close()

    def fold(self, w1: int, w2: int) -> "FoldedGens":
        if self.off or self.length != len(self.root) >> len(self.w1s) or self.length % 2:
            raise ValueError("fold needs the whole, even-sized set of folded generators")
        return FoldedGens(self.root, self.w1s + [w1], self.w2s + [w2])

    def split_at(self, k: int) -> Tuple["FoldedGens", "FoldedGens"]:
        return FoldedGens(self.root, self.w1s, self.w2s, self.off, k), FoldedGens(self.root, self.w1s, self.w2s, self.off + k, self.length - k)

    def materialize(self) -> "CommitmentGens":
        bases = self.root.bases
        for w1, w2 in zip(self.w1s, self.w2s):
            bases = msm.fold(self.curve, bases, bases.shape[0] // 2, w1, w2)
        return CommitmentGens(self.curve, bases[self.off:self.off + self.length].copy(), self.h, precompute=False)


class HyraxPC:
    """nova-snark `HyraxPC` as Reef builds it at src/backend/commitment.rs:182-185: row generators
    `gens_v` (2^right of them) with the blinding generator of `gens_s`."""

    def __init__(self, gens_v: CommitmentGens):
        if gens_v.h is None:
            raise ValueError("HyraxPC needs a blinding generator")
        self.gens_v = gens_v

    @staticmethod
    def compute_factored_lens(num_vars: int) -> Tuple[int, int]:
        """EqPolynomial::compute_factored_lens [R]: (left, right) = (l/2, l - l/2)."""
        return num_vars // 2, num_vars - num_vars // 2

    def commit(self, poly: np.ndarray, blinds: np.ndarray, *, is_mont: bool = True, max_scalar_bits: int = 0):
        """HyraxPC::commit(&poly): the 2^l evaluations are viewed as an L x R matrix (L = 2^left
        rows); returns the L row commitments  sum_j Z[i, j] * G_j + blinds[i] * H  and their
        32-byte compressed forms (what Reef absorbs into the Poseidon RO, commitment.rs:190-198)."""
        poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
        n = poly.shape[0]
        if n & (n - 1):
            raise ValueError("polynomial length must be a power of two")
        left, right = self.compute_factored_lens(n.bit_length() - 1)
        rows, row_len = 1 << left, 1 << right
        if row_len > len(self.gens_v):
            raise ValueError("not enough row generators")
        blinds = np.ascontiguousarray(blinds, dtype=np.uint64).reshape(rows, 4)
        target = self.gens_v._on_devices() or self.gens_v._context()        # rows dealt out whole over the devices, or one context
        out = target.msm_rows(poly, rows, row_len, is_mont=is_mont, max_scalar_bits=max_scalar_bits, blinds=blinds, h=self.gens_v.h)
        comp = msm.normalize(self.gens_v.curve, out, affine=False, compressed=True)[1]
        return out, comp

    def commit_symbols(self, symbols: np.ndarray, blinds: np.ndarray, symbol_bits: int, *, blinds_are_mont: bool = True):
        """HyraxPC::commit from the document symbols themselves (uint8, < 2^symbol_bits; the values Reef turns into
        the polynomial's evaluations, framework.rs:978-1011): same row commitments as `commit`."""
        symbols = np.ascontiguousarray(symbols, dtype=np.uint8).reshape(-1)
        n = symbols.shape[0]
        if n & (n - 1):
            raise ValueError("polynomial length must be a power of two")
        left, right = self.compute_factored_lens(n.bit_length() - 1)
        rows, row_len = 1 << left, 1 << right
        if row_len > len(self.gens_v):
            raise ValueError("not enough row generators")
        blinds = np.ascontiguousarray(blinds, dtype=np.uint64).reshape(rows, 4)
        target = self.gens_v._on_devices() or self.gens_v._context()
        out = target.msm_rows_symbols(symbols, rows, row_len, symbol_bits, blinds=blinds, h=self.gens_v.h, blinds_are_mont=blinds_are_mont)
        comp = msm.normalize(self.gens_v.curve, out, affine=False, compressed=True)[1]
        return out, comp

    def bind_rows(self, poly, blinds: np.ndarray, point: np.ndarray, *, is_mont: bool = True, n=None, elem_bytes=None):
        """First step of HyraxPC::prove_eval (src/backend/commitment.rs:371-391): with (L, R) the
        eq-evaluations of the two halves of `point`, returns (LZ = L^T Z, eval = <LZ, R>,
        LZ_blind = <L, blinds>) in the form `is_mont` names.  `poly` is an (n, 4) uint64 array of
        field elements, a uint8/16/32 symbol array, or a device buffer (then give n, elem_bytes);
        the dot-product IPA over LZ that follows uses `CommitmentGens.commit` / `ipa_cross_terms`."""
        from . import mle
        point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
        left, _ = self.compute_factored_lens(point.shape[0])
        lz, ev = mle.bound_rows_raw(self.gens_v.curve, poly, point, left, is_mont=is_mont, n=n, elem_bytes=elem_bytes)
        blinds = np.ascontiguousarray(blinds, dtype=np.uint64).reshape(1 << left, 4)
        lz_blind, _ = mle.bound_rows_raw(self.gens_v.curve, blinds, point[:left], left, is_mont=is_mont)
        return lz, ev, lz_blind[0]

