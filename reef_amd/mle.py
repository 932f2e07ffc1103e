"""Python front-end of row N3: bound rows / evaluation of a multilinear table on the GPU.

Mirrors what NLDocCommitment::proof_dot_prod_prover (src/backend/commitment.rs:287-405) asks of the
document polynomial: `doc_poly.evaluate(&running_q)` (:357) and the row binding `LZ = L^T Z` at the
top of `hyrax_gen.prove_eval` (:371-391); also `verifier_mle_eval(table, q')` (:236).  All
arithmetic runs in libreef_msm.so (reef_mle_bound_rows).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _ffi
from ._ffi import REEF_DEVICE, REEF_HOST, check
from .msm import DeviceBuffer, curve_id
from .sumcheck import array_to_ints, ints_to_array

Table = Union[np.ndarray, DeviceBuffer, int]


def _table_ptr(z: Table, n: Optional[int], elem_bytes: Optional[int]) -> Tuple[int, int, int, int]:
    """-> (loc, ptr, n, elem_bytes)."""
    if isinstance(z, np.ndarray):
        if not z.flags["C_CONTIGUOUS"]:
            raise ValueError("host tables must be C-contiguous")
        if z.dtype == np.uint64:                       # (n, 4) field elements
            zz = z.reshape(-1, 4)
            return REEF_HOST, zz.ctypes.data, zz.shape[0], 32
        eb = {np.dtype(np.uint8): 1, np.dtype(np.uint16): 2, np.dtype(np.uint32): 4}.get(z.dtype)
        if eb is None:
            raise TypeError("tables are uint64 (n,4) field elements or uint8/uint16/uint32 symbols")
        return REEF_HOST, z.ctypes.data, z.size, eb
    if n is None or elem_bytes is None:
        raise ValueError("n and elem_bytes are required for device-resident tables")
    ptr = z.ptr if isinstance(z, DeviceBuffer) else int(z)
    if isinstance(z, DeviceBuffer) and z.nbytes < n * elem_bytes:
        raise ValueError("device buffer too small")
    return REEF_DEVICE, ptr, n, elem_bytes


def bound_rows_raw(curve, z: Table, point: np.ndarray, left_vars: int, *, is_mont: bool = True, n: Optional[int] = None,
                   elem_bytes: Optional[int] = None, want_rows: bool = True) -> Tuple[Optional[np.ndarray], np.ndarray]:
    """ABI-level call: `point` is an (m, 4) uint64 array in the form `is_mont` names; returns
    (LZ as a (2^(m-left), 4) array or None, eval as a (4,) array) in the same form."""
    lib = _ffi.load()
    point = np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 4)
    m = point.shape[0]
    loc, ptr, n, eb = _table_ptr(z, n, elem_bytes)
    lz = np.zeros((1 << max(0, m - left_vars), 4), dtype=np.uint64) if want_rows else None   # bad shapes: the library reports them
    ev = np.zeros(4, dtype=np.uint64)
    check(lib.reef_mle_bound_rows(curve_id(curve), ptr, n, eb, loc, bool(is_mont), point.ctypes.data if m else None, m, left_vars,
                                  lz.ctypes.data if want_rows else None, REEF_HOST, ev.ctypes.data))
    return lz, ev


def bound_rows(curve, z: Sequence[int] | np.ndarray, point: Sequence[int], left_vars: Optional[int] = None) -> Tuple[List[int], int]:
    """Canonical-integer convenience form (the reference's rug::Integer view): z is a list of ints
    (sent as 32-byte elements) or a uint8/16/32 symbol array; returns (LZ, eval) as Python ints."""
    m = len(point)
    if left_vars is None:
        left_vars = m // 2          # compute_factored_lens, commitment.rs:173-174
    table = z if isinstance(z, np.ndarray) and z.dtype != np.uint64 else ints_to_array(list(z))
    lz, ev = bound_rows_raw(curve, table, ints_to_array(list(point)), left_vars, is_mont=False)
    return array_to_ints(lz), array_to_ints(ev)[0]


def evaluate(curve, z, point: Sequence[int]) -> int:
    """doc_poly.evaluate(point) (commitment.rs:357) / verifier_mle_eval(table, point) (:236)."""
    return bound_rows(curve, z, point)[1]


def verifier_mle_eval(table, q: Sequence[int], curve="pallas") -> int:
    """The reference's verifier_mle_eval(table, q) (src/backend/r1cs_helper.rs:637-641; called at commitment.rs:236 on the document): the
    multilinear extension of `table` at q, q[0] pairing with the most significant index bit.  Over Fq = the scalar field of Pallas, as Reef's."""
    return evaluate(curve, table, q)
