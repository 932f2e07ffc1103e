"""Row N1: commitment-key derivation (nova-snark's CommitmentGens::new -> from_label [R], called at
src/backend/framework.rs:297-303 and src/backend/commitment.rs:146-149,176-180) through the C ABI
(reef_derive_generators).  Everything pasta_curves' hash_to_curve fixes -- E', Z, the isogeny, the domain separation
string, the byte order -- is the caller's data; this module only marshals it."""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from ._ffi import REEF_DEVICE, REEF_HOST, check
from .msm import curve_id

Point = Optional[Tuple[int, int]]
BASE_MODULUS = {0: 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001,      # Pallas: coordinates in Fp
                1: 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001}      # Vesta: coordinates in Fq


def _fe(v: int):
    return (ctypes.c_uint64 * 4)(*[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)])


class KeygenParams(ctypes.Structure):
    _fields_ = [("a", ctypes.c_uint64 * 4), ("b", ctypes.c_uint64 * 4), ("z", ctypes.c_uint64 * 4), ("iso", (ctypes.c_uint64 * 4) * 13),
                ("dst", ctypes.c_char_p), ("dst_len", ctypes.c_uint32), ("little_endian", ctypes.c_uint32)]


def shake256(data: bytes, out_len: int) -> bytes:
    """The library's own SHAKE256 (the host half of the derivation)."""
    out = ctypes.create_string_buffer(out_len)
    _ffi.load().reef_shake256(data, len(data), out, out_len)
    return out.raw


def derive_generators(curve, label: bytes, n: int, a: int, b: int, z: int, iso: Sequence[int], dst: bytes,
                      little_endian: bool = False, device: bool = False):
    """-> (n, 8) uint64: n affine points in the ABI form (what reef_msm_ctx_create takes as a key), or with `device` a
    DeviceBuffer holding them (the key never visits the host).  Parameters as canonical integers of the curve's base field."""
    assert len(iso) == 13
    lib = _ffi.load()
    kp = KeygenParams(_fe(a), _fe(b), _fe(z), ((ctypes.c_uint64 * 4) * 13)(*[_fe(c) for c in iso]), dst, len(dst), 1 if little_endian else 0)
    if device:
        from .msm import DeviceBuffer
        buf = DeviceBuffer(64 * n)
        check(lib.reef_derive_generators(curve_id(curve), label, len(label), n, ctypes.byref(kp), False, buf.ptr, REEF_DEVICE))
        return buf
    out = np.zeros((n, 8), dtype=np.uint64)
    check(lib.reef_derive_generators(curve_id(curve), label, len(label), n, ctypes.byref(kp), False, out.ctypes.data, REEF_HOST))
    return out


def points_to_ints(curve, raw: np.ndarray) -> List[Point]:
    """ABI affine points (pasta Montgomery form, R = 2^256) -> canonical integer pairs."""
    p = BASE_MODULUS[curve_id(curve)]
    rinv = pow(1 << 256, -1, p)
    res: List[Point] = []
    for row in np.asarray(raw, dtype=np.uint64).reshape(-1, 8):
        x = sum(int(row[j]) << (64 * j) for j in range(4))
        y = sum(int(row[4 + j]) << (64 * j) for j in range(4))
        res.append(None if x == 0 and y == 0 else (x * rinv % p, y * rinv % p))
    return res
