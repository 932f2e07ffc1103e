"""ctypes front-end of oracle/libpasta_ref.so (the C restatement).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
All buffers are numpy uint64 arrays in the C-ABI layouts (see pasta_ref.h).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpasta_ref.so")

PALLAS, VESTA = 0, 1


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pasta_ref.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpasta_ref.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        vp, sz, i32, u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64
        _lib.pasta_ref_msm_naive.argtypes = [i32, vp, vp, sz, i32, vp]
        _lib.pasta_ref_msm_pippenger.argtypes = [i32, vp, vp, sz, i32, i32, vp]
        _lib.pasta_ref_msm_pippenger_windows.argtypes = [i32, vp, vp, sz, i32, i32, vp]
        _lib.pasta_ref_window_plan.argtypes = [sz, i32, ctypes.POINTER(ctypes.c_uint)]
        _lib.pasta_ref_window_plan.restype = ctypes.c_uint
        _lib.pasta_ref_fold_mt.argtypes = [i32, vp, sz, vp, vp, i32, vp]
        _lib.pasta_ref_to_affine.argtypes = [i32, vp, sz, vp]
        _lib.pasta_ref_compress.argtypes = [i32, vp, sz, vp]
        _lib.pasta_ref_scalar_mul.argtypes = [i32, vp, vp, vp]
        _lib.pasta_ref_gen_bases_ap.argtypes = [i32, u64, u64, sz, vp]
        _lib.pasta_ref_gen_scalars.argtypes = [i32, u64, i32, u64, sz, i32, vp]
        for name in ("fmul", "fadd", "fsub"):
            getattr(_lib, "pasta_ref_" + name).argtypes = [i32, vp, vp, vp]
        for name in ("finv", "to_mont", "from_mont"):
            getattr(_lib, "pasta_ref_" + name).argtypes = [i32, vp, vp]
        _lib.pasta_ref_fold.argtypes = [i32, vp, sz, vp, vp, vp]
        _lib.pasta_ref_row_msm.argtypes = [i32, vp, vp, vp, vp, sz, sz, i32, i32, vp]
        _lib.pasta_ref_sc_to_mont.argtypes = [i32, vp, sz]
        _lib.pasta_ref_sc_from_mont.argtypes = [i32, vp, sz]
        _lib.pasta_ref_sc_round.argtypes = [i32, vp, vp, sz, vp, vp]
    return _lib


def _p(a: np.ndarray) -> int:
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def gen_bases_ap(curve: int, k0: int, d: int, n: int) -> np.ndarray:
    out = np.zeros((n, 8), dtype=np.uint64)
    lib().pasta_ref_gen_bases_ap(curve, k0, d, n, _p(out))
    return out


def gen_scalars(curve: int, seed: int, n: int, kind: int = 0, small_bound: int = 0, mont: bool = True) -> np.ndarray:
    out = np.zeros((n, 4), dtype=np.uint64)
    lib().pasta_ref_gen_scalars(curve, seed, kind, small_bound, n, int(mont), _p(out))
    return out


def msm_naive(curve: int, bases: np.ndarray, scalars: np.ndarray, mont: bool = True) -> np.ndarray:
    out = np.zeros(12, dtype=np.uint64)
    lib().pasta_ref_msm_naive(curve, _p(bases), _p(scalars), bases.shape[0], int(mont), _p(out))
    return out


def msm_pippenger(curve: int, bases: np.ndarray, scalars: np.ndarray, mont: bool = True, threads: int = 1) -> np.ndarray:
    out = np.zeros(12, dtype=np.uint64)
    lib().pasta_ref_msm_pippenger(curve, _p(bases), _p(scalars), bases.shape[0], int(mont), threads, _p(out))
    return out


def msm_pippenger_windows(curve: int, bases: np.ndarray, scalars: np.ndarray, mont: bool = True, threads: int = 1, n: int | None = None) -> np.ndarray:
    """Window-parallel Pippenger on the persistent thread pool (the timed cpu_baseline); n: prefix length."""
    out = np.zeros(12, dtype=np.uint64)
    lib().pasta_ref_msm_pippenger_windows(curve, _p(bases), _p(scalars), bases.shape[0] if n is None else n, int(mont), threads, _p(out))
    return out


def pool_size() -> int:
    return int(lib().pasta_ref_pool_size())


def window_plan(n: int, threads: int) -> tuple[int, int]:
    s = ctypes.c_uint(0)
    c = lib().pasta_ref_window_plan(n, threads, ctypes.byref(s))
    return int(c), int(s.value)


def to_affine(curve: int, jac: np.ndarray) -> np.ndarray:
    jac = np.ascontiguousarray(jac.reshape(-1, 12))
    out = np.zeros((jac.shape[0], 8), dtype=np.uint64)
    lib().pasta_ref_to_affine(curve, _p(jac), jac.shape[0], _p(out))
    return out


def compress(curve: int, jac: np.ndarray) -> bytes:
    jac = np.ascontiguousarray(jac.reshape(-1, 12))
    out = np.zeros(32 * jac.shape[0], dtype=np.uint8)
    lib().pasta_ref_compress(curve, _p(jac), jac.shape[0], _p(out))
    return out.tobytes()


def scalar_mul(curve: int, base: np.ndarray, k: int) -> np.ndarray:
    kk = np.array([(k >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    out = np.zeros(12, dtype=np.uint64)
    lib().pasta_ref_scalar_mul(curve, _p(np.ascontiguousarray(base)), _p(kk), _p(out))
    return out


def fold(curve: int, gens: np.ndarray, w1: int, w2: int) -> np.ndarray:
    half = gens.shape[0] // 2
    a = np.array([(w1 >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    b = np.array([(w2 >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    out = np.zeros((half, 8), dtype=np.uint64)
    lib().pasta_ref_fold(curve, _p(gens), half, _p(a), _p(b), _p(out))
    return out


def fold_mt(curve: int, gens: np.ndarray, w1: int, w2: int, threads: int, half: int | None = None) -> np.ndarray:
    half = gens.shape[0] // 2 if half is None else half
    a, b = int_to_limbs(w1), int_to_limbs(w2)
    out = np.zeros((half, 8), dtype=np.uint64)
    lib().pasta_ref_fold_mt(curve, _p(gens), half, _p(a), _p(b), threads, _p(out))
    return out


def row_msm(curve: int, bases: np.ndarray, scalars: np.ndarray, rows: int, row_len: int, h=None, blinds=None,
            mont: bool = True, threads: int = 1) -> np.ndarray:
    out = np.zeros((rows, 12), dtype=np.uint64)
    lib().pasta_ref_row_msm(curve, _p(bases), _p(h) if h is not None else None, _p(scalars),
                            _p(blinds) if blinds is not None else None, rows, row_len, int(mont), threads, _p(out))
    return out


def field_op(name: str, field: int, a: np.ndarray, b: np.ndarray | None = None) -> np.ndarray:
    out = np.zeros(4, dtype=np.uint64)
    fn = getattr(lib(), "pasta_ref_" + name)
    if b is None:
        fn(field, _p(a), _p(out))
    else:
        fn(field, _p(a), _p(b), _p(out))
    return out


def int_to_limbs(v: int) -> np.ndarray:
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def limbs_to_int(a) -> int:
    return sum(int(x) << (64 * i) for i, x in enumerate(np.asarray(a).reshape(-1)[:4]))


def sc_round(field: int, T: np.ndarray, E: np.ndarray, pow_: int, r: int) -> tuple:
    """One sum-check round on Montgomery-form tables (see sc_to_mont); returns (xsq, x, con)."""
    out = np.zeros((3, 4), dtype=np.uint64)
    rr = int_to_limbs(r)
    lib().pasta_ref_sc_round(field, _p(T), _p(E), pow_, _p(rr), _p(out))
    return tuple(limbs_to_int(out[i]) for i in range(3))


def sc_to_mont(field: int, table: np.ndarray) -> None:
    lib().pasta_ref_sc_to_mont(field, _p(table), table.shape[0])


def sc_from_mont(field: int, table: np.ndarray) -> None:
    lib().pasta_ref_sc_from_mont(field, _p(table), table.shape[0])
