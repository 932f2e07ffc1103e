"""Python front-end of the sum-check vector kernels (row N2 of SURVEY.md 8f).

Mirrors the host helpers of src/backend/r1cs_helper.rs as Reef's witness generator drives them
(src/backend/r1cs.rs:2318-2385): `gen_eq_table`, then per round `linear_mle_product` split into its
two halves around the Poseidon challenge, which stays on the host.  Values are canonical Python
ints (the reference uses rug::Integer); all arithmetic runs in libreef_msm.so.
"""
from __future__ import annotations

import ctypes
from typing import List, Sequence, Tuple

import numpy as np

from . import _ffi
from ._ffi import REEF_DEVICE, REEF_HOST, check
from .msm import curve_id


def ints_to_array(values: Sequence[int]) -> np.ndarray:
    out = np.zeros((len(values), 4), dtype=np.uint64)
    for i, v in enumerate(values):
        for j in range(4):
            out[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def array_to_ints(arr: np.ndarray) -> List[int]:
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [sum(int(arr[i, j]) << (64 * j) for j in range(4)) for i in range(arr.shape[0])]


class SumCheck:
    """Resident (T, EQ) tables of 2^ell entries over the scalar field of `curve`."""

    def __init__(self, curve, ell: int):
        self._lib = _ffi.load()
        self.ell = ell
        self.len = 1 << ell
        h = ctypes.c_void_p()
        check(self._lib.reef_sc_create(ctypes.byref(h), curve_id(curve), self.len))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.reef_sc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_table(self, which: int, values) -> None:
        """values: list of ints, or an (n, 4) uint64 array of canonical limbs."""
        arr = values if isinstance(values, np.ndarray) else ints_to_array(values)
        arr = np.ascontiguousarray(arr, dtype=np.uint64)
        self._pending = None
        check(self._lib.reef_sc_set_table(self._h, which, arr.ctypes.data, arr.shape[0], REEF_HOST))

    def set_table_device(self, which: int, ptr: int, n: int) -> None:
        self._pending = None
        check(self._lib.reef_sc_set_table(self._h, which, ptr, n, REEF_DEVICE))

    def gen_eq_table(self, rs: Sequence[int], qs: Sequence[int], last_q: Sequence[int]) -> None:
        """gen_eq_table(rs, qs, last_q) (r1cs_helper.rs:508-544) into the EQ table."""
        assert len(rs) == len(qs) + 1 and len(last_q) == self.ell
        a, lq = ints_to_array(rs), ints_to_array(last_q)
        q = np.ascontiguousarray(np.array(list(qs) + [0], dtype=np.uint32))
        self._pending = None
        check(self._lib.reef_sc_gen_eq_table(self._h, a.ctypes.data, q.ctypes.data, len(qs), lq.ctypes.data, self.ell))

    def round_coeffs(self, i: int) -> Tuple[int, int, int]:
        """First half of linear_mle_product for round i in 1..ell: (xsq, x, con)."""
        out = np.zeros((3, 4), dtype=np.uint64)
        check(self._lib.reef_sc_round_coeffs(self._h, 1 << (self.ell - i), out.ctypes.data))
        xsq, x, con = array_to_ints(out)
        return xsq, x, con

    def fold(self, i: int, r: int) -> None:
        """Second half of linear_mle_product for round i with the host's challenge r."""
        rr = ints_to_array([r])
        self._pending = None
        check(self._lib.reef_sc_fold(self._h, 1 << (self.ell - i), rr.ctypes.data))

    def fold_and_next_coeffs(self, i: int, r: int) -> Tuple[int, int, int]:
        """fold(i, r) fused with round_coeffs(i + 1): one pass over the tables (i < ell)."""
        rr = ints_to_array([r])
        out = np.zeros((3, 4), dtype=np.uint64)
        self._pending = None
        check(self._lib.reef_sc_fold_and_next_coeffs(self._h, 1 << (self.ell - i), rr.ctypes.data, out.ctypes.data))
        xsq, x, con = array_to_ints(out)
        return xsq, x, con

    def linear_mle_product(self, i: int, sponge) -> Tuple[int, int, int, int]:
        """The whole of the reference's linear_mle_product(table_t, table_eq, ell, i, sponge) (src/backend/r1cs_helper.rs:441-506) on the
        resident tables: the round's sums, absorb (con, x, xsq) in that order (:478-482), squeeze the challenge (:485-488), fold both
        tables.  Returns (r_i, xsq, x, con) like the reference.  `sponge` is the caller's transcript (absorb(list of ints), squeeze(n) ->
        list: the SpongeAPI calls Reef makes; neptune's on the Rust side): the challenge is the host's.  Rounds driven one after the
        other as r1cs.rs:2318-2385 does cost one pass over the tables each: the fold of round i also yields round i + 1's sums."""
        pending = getattr(self, "_pending", None)
        xsq, x, con = pending[1] if pending is not None and pending[0] == i else self.round_coeffs(i)
        sponge.absorb([con, x, xsq])
        r_i = sponge.squeeze(1)[0]
        if i < self.ell:
            self._pending = (i + 1, self.fold_and_next_coeffs(i, r_i))
        else:
            self.fold(i, r_i)
            self._pending = None
        return r_i, xsq, x, con

    def read(self, which: int, count: int) -> List[int]:
        out = np.zeros((count, 4), dtype=np.uint64)
        check(self._lib.reef_sc_read(self._h, which, count, out.ctypes.data))
        return array_to_ints(out)

    def reset_table(self) -> None:
        """T <- the table as last set (start of the next folding step)."""
        self._pending = None
        check(self._lib.reef_sc_reset_table(self._h))

    def sync(self) -> None:
        check(self._lib.reef_sc_sync(self._h))
