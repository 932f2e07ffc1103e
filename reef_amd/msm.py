"""Python front-end of the C ABI: resident keys, MSMs, row commitments, folds, normalisation.

All arithmetic happens in libreef_msm.so on the GPU; this module only marshals buffers.
Host buffers are numpy uint64 arrays in the ABI layouts:
    scalars (n, 4)   affine points (n, 8)   jacobian points (n, 12)
Device buffers are `DeviceBuffer` objects (or raw integer device pointers, e.g.
`torch_tensor.data_ptr()`).
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple, Union

import numpy as np

from . import _ffi
from ._ffi import PALLAS, REEF_DEVICE, REEF_HOST, VESTA, MsmOpts, ReefError, check  # noqa: F401

CURVE_IDS = {"pallas": PALLAS, "vesta": VESTA, PALLAS: PALLAS, VESTA: VESTA}
# The two Pasta primes (the same constants as csrc/field_consts.h): Pallas is y^2 = x^3 + 5 over F_p with q points, Vesta the
# same equation over F_q with p points.  Reef's CirC modulus is q (src/backend/r1cs_helper.rs:37-38).
PALLAS_BASE_P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001
PALLAS_SCALAR_Q = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
SCALAR_MODULUS = {PALLAS: PALLAS_SCALAR_Q, VESTA: PALLAS_BASE_P}
Buf = Union[np.ndarray, "DeviceBuffer", int]


def curve_id(curve) -> int:
    try:
        return CURVE_IDS[curve]
    except KeyError:
        raise ValueError(f"unknown curve {curve!r}") from None


class DeviceBuffer:
    """hipMalloc'ed memory owned by Python."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        self.ptr = _ffi.load().reef_device_alloc(self.nbytes)
        if not self.ptr:
            raise ReefError(4, _ffi.load().reef_last_error().decode())

    @classmethod
    def from_host(cls, arr: np.ndarray) -> "DeviceBuffer":
        arr = np.ascontiguousarray(arr)
        buf = cls(arr.nbytes)
        check(_ffi.load().reef_memcpy(buf.ptr, arr.ctypes.data, arr.nbytes, REEF_DEVICE, REEF_HOST))
        return buf

    def to_host(self, shape, dtype=np.uint64) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        check(_ffi.load().reef_memcpy(out.ctypes.data, self.ptr, out.nbytes, REEF_HOST, REEF_DEVICE))
        return out

    def free(self) -> None:
        if self.ptr:
            _ffi.load().reef_device_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _loc_ptr(buf: Buf, min_bytes: int = 0) -> Tuple[int, int]:
    """-> (location flag, raw pointer)."""
    if isinstance(buf, DeviceBuffer):
        if buf.nbytes < min_bytes:
            raise ValueError("device buffer too small")
        return REEF_DEVICE, buf.ptr
    if isinstance(buf, np.ndarray):
        if buf.dtype != np.uint64 and buf.dtype != np.uint8:
            raise TypeError("host buffers must be uint64 (or uint8) numpy arrays")
        if not buf.flags["C_CONTIGUOUS"]:
            raise ValueError("host buffers must be C-contiguous")
        if buf.nbytes < min_bytes:
            raise ValueError(f"host buffer too small: {buf.nbytes} < {min_bytes} bytes")
        return REEF_HOST, buf.ctypes.data
    if isinstance(buf, int):
        return REEF_DEVICE, buf
    raise TypeError(f"unsupported buffer type {type(buf)}")


def scalar_to_limbs(v: int) -> np.ndarray:
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


class MsmContext:
    """A resident commitment key (nova-snark `CommitmentGens<G>`) on one GPU."""

    def __init__(self, curve, bases: Buf, n: Optional[int] = None, *, window_bits: int = 0, bucket_groups: int = 0,
                 chunk: int = 0, device: int = -1, byte_tables: int = 0, _handle=None):
        self.curve = curve_id(curve)
        self._lib = _ffi.load()
        if _handle is not None:
            self._h = _handle
            self.n = n
            return
        if n is None:
            if not isinstance(bases, np.ndarray):
                raise ValueError("n is required for device-resident bases")
            n = bases.shape[0]
        loc, ptr = _loc_ptr(bases, 64 * n)
        opts = MsmOpts(window_bits, bucket_groups, chunk, byte_tables, device, (ctypes.c_uint32 * 3)(0, 0, 0))
        h = ctypes.c_void_p()
        check(self._lib.reef_msm_ctx_create(ctypes.byref(h), self.curve, ptr, n, loc, ctypes.byref(opts)))
        self._h = h
        self.n = n

    def set_bases(self, bases: Buf, n: Optional[int] = None) -> None:
        """Replace the key in place (IPA rounds: the generators change every round)."""
        if n is None:
            n = bases.shape[0]
        loc, ptr = _loc_ptr(bases, 64 * n)
        check(self._lib.reef_msm_ctx_set_bases(self._h, ptr, n, loc))
        self.n = n

    def clone(self) -> "MsmContext":
        h = ctypes.c_void_p()
        check(self._lib.reef_msm_ctx_clone(ctypes.byref(h), self._h))
        return MsmContext(self.curve, None, self.n, _handle=h)

    def attach(self, other: "MsmContext") -> None:
        """This handle moves to `other`'s resident key (as a clone of it would be), keeping its stream and workspace."""
        check(self._lib.reef_msm_ctx_attach(self._h, other._h))
        self.n = other.n

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.reef_msm_ctx_destroy(self._h)
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

    def sync(self) -> None:
        check(self._lib.reef_msm_ctx_sync(self._h))

    @property
    def stream(self) -> int:
        return self._lib.reef_msm_ctx_stream(self._h) or 0

    def plan(self) -> dict:
        c, w, g, t = (ctypes.c_uint32() for _ in range(4))
        check(self._lib.reef_msm_ctx_plan(self._h, ctypes.byref(c), ctypes.byref(w), ctypes.byref(g), ctypes.byref(t)))
        return {"window_bits": c.value, "windows": w.value, "bucket_groups": g.value, "tables": t.value}

    def has_byte_tables(self) -> bool:
        """True when this key's MSMs are served from its byte tables (opt-in: byte_tables = 1 at creation, 3 in the background; reef_msm.h)."""
        return bool(self._lib.reef_msm_ctx_byte_tables(self._h))

    def set_window_split(self, rank: int, world: int) -> None:
        """This context accumulates only the windows w = rank (mod world): its MSMs return partial sums."""
        check(self._lib.reef_msm_ctx_set_window_split(self._h, rank, world))

    def enable_timing(self, on: bool = True) -> None:
        """HIP-event timing of every MSM on this context (off by default: ~6 us per event)."""
        check(self._lib.reef_msm_ctx_enable_timing(self._h, int(on)))

    def last_timing(self) -> Tuple[float, float]:
        """(total ms, accumulation-kernel ms) of the last MSM on this context (HIP events)."""
        a, b = ctypes.c_float(), ctypes.c_float()
        check(self._lib.reef_msm_ctx_last_timing(self._h, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def timing_stats(self, reset: bool = True) -> dict:
        """Sums over the MSMs issued since the last reset (waits for those still in flight)."""
        calls, tot, acc = ctypes.c_uint64(), ctypes.c_double(), ctypes.c_double()
        check(self._lib.reef_msm_ctx_timing_stats(self._h, int(reset), ctypes.byref(calls), ctypes.byref(tot), ctypes.byref(acc)))
        return {"calls": calls.value, "total_ms": tot.value, "accumulate_ms": acc.value}

    def sum_points(self, jac: Buf, n: int, out: Buf) -> Buf:
        """Sum n device-resident Jacobian points on this context's stream."""
        loc, ptr = _loc_ptr(jac, 96 * n)
        oloc, optr = _loc_ptr(out, 96)
        if loc != REEF_DEVICE or oloc != REEF_DEVICE:
            raise ValueError("device buffers only")
        check(self._lib.reef_msm_ctx_sum_points(self._h, ptr, n, optr))
        return out

    def msm(self, scalars: Buf, n: Optional[int] = None, *, is_mont: bool = True, out: Optional[Buf] = None) -> Buf:
        """sum_i scalars[i]*bases[i].  Host result: uint64[12] Jacobian.  With a device `out`
        the call only enqueues work on the context's stream."""
        if n is None:
            if not isinstance(scalars, np.ndarray):
                raise ValueError("n is required for device-resident scalars")
            n = scalars.shape[0]
        if n > self.n:
            raise ValueError(f"n = {n} exceeds the key length {self.n}")  # length mismatch panics in the reference
        loc, ptr = _loc_ptr(scalars, 32 * n)
        if out is None:
            out = np.zeros(12, dtype=np.uint64)
        oloc, optr = _loc_ptr(out, 96)
        check(self._lib.reef_msm(self._h, ptr, n, loc, bool(is_mont), optr, oloc))
        return out

    def msm_rows_symbols(self, symbols: Buf, rows: int, row_len: int, symbol_bits: int, *, blinds: Optional[Buf] = None,
                         h: Optional[Buf] = None, blinds_are_mont: bool = True, out: Optional[Buf] = None) -> Buf:
        """HyraxPC::commit on one-byte document symbols (< 2^symbol_bits): uint8 array or device buffer of rows*row_len bytes."""
        if isinstance(symbols, np.ndarray) and symbols.dtype != np.uint8:
            raise TypeError("symbols must be uint8")
        loc, ptr = _loc_ptr(symbols, rows * row_len)
        bp = hp = None
        if blinds is not None:
            if h is None:
                raise ValueError("blinds need the blinding generator h")
            bloc, bp = _loc_ptr(blinds, 32 * rows)
            hloc, hp = _loc_ptr(h, 64)
            if bloc != loc or hloc != loc:
                raise ValueError("blinds and h must live where the symbols live")
        if out is None:
            out = np.zeros((rows, 12), dtype=np.uint64)
        oloc, optr = _loc_ptr(out, 96 * rows)
        check(self._lib.reef_msm_rows_symbols(self._h, ptr, rows, row_len, loc, symbol_bits, bp, hp, bool(blinds_are_mont), optr, oloc))
        return out

    def ipa_cross_terms(self, a: np.ndarray, w1s, w2s, *, is_mont: bool = True):
        """Cross terms (L, R) of IPA round k = len(w1s) over the original generators, without
        folding them: a = a_lo || a_hi (n / 2^k scalars), w1s/w2s = challenges so far (ints)."""
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
        k = len(w1s)
        assert len(w2s) == k
        w1 = np.array([list(scalar_to_limbs(w)) for w in w1s], dtype=np.uint64).reshape(k, 4) if k else np.zeros((1, 4), np.uint64)
        w2 = np.array([list(scalar_to_limbs(w)) for w in w2s], dtype=np.uint64).reshape(k, 4) if k else np.zeros((1, 4), np.uint64)
        out_l, out_r = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64)
        check(self._lib.reef_ipa_cross_terms(self._h, a.ctypes.data, a.shape[0], REEF_HOST, bool(is_mont), w1.ctypes.data,
                                             w2.ctypes.data, k, out_l.ctypes.data, out_r.ctypes.data))
        return out_l, out_r

    def msm_folded(self, v: np.ndarray, w1s, w2s, off: int = 0, *, is_mont: bool = True) -> np.ndarray:
        """Commitment over the generators len(w1s) folds away from this key, slice [off, off + len(v)), without folding
        them (reef_msm_folded): v (len, 4) uint64 scalars, challenges as ints."""
        v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
        k = len(w1s)
        assert len(w2s) == k
        w1 = np.array([list(scalar_to_limbs(w)) for w in w1s], dtype=np.uint64).reshape(k, 4) if k else np.zeros((1, 4), np.uint64)
        w2 = np.array([list(scalar_to_limbs(w)) for w in w2s], dtype=np.uint64).reshape(k, 4) if k else np.zeros((1, 4), np.uint64)
        out = np.zeros(12, dtype=np.uint64)
        check(self._lib.reef_msm_folded(self._h, v.ctypes.data, v.shape[0], off, REEF_HOST, bool(is_mont), w1.ctypes.data, w2.ctypes.data, k,
                                        out.ctypes.data, REEF_HOST))
        return out

    def msm_rows(self, scalars: Buf, rows: int, row_len: int, *, is_mont: bool = True, max_scalar_bits: int = 0,
                 blinds: Optional[Buf] = None, h: Optional[Buf] = None, out: Optional[Buf] = None) -> Buf:
        if row_len > self.n:
            raise ValueError(f"row_len = {row_len} exceeds the key length {self.n}")
        if (blinds is None) != (h is None):
            raise ValueError("blinds and h go together")
        loc, ptr = _loc_ptr(scalars, 32 * rows * row_len)
        bptr = hptr = None
        if blinds is not None:
            bloc, bptr = _loc_ptr(blinds, 32 * rows)
            hloc, hptr = _loc_ptr(h, 64)
            if bloc != loc or hloc != loc:
                raise ValueError("blinds and h must live where the scalars live")
        if out is None:
            out = np.zeros((rows, 12), dtype=np.uint64)
        oloc, optr = _loc_ptr(out, 96 * rows)
        check(self._lib.reef_msm_rows(self._h, ptr, rows, row_len, loc, bool(is_mont), max_scalar_bits, bptr, hptr, optr, oloc))
        return out


SPLIT_WINDOWS, SPLIT_POINTS = 0, 1
SCALARS_EACH, SCALARS_FANOUT = 0, 1      # reef_msm_group_opts.scalars: how host scalars reach the members of a windows group
EXCHANGE_PEER, EXCHANGE_HOST, EXCHANGE_RCCL = 1, 2, 3


class MsmGroup:
    """A commitment key resident on several GPUs of THIS process (reef_msm_group_*, include/reef_msm.h section 5): one MSM split by
    Pippenger window (every device holds the key) or by points (every device holds a slice), the 96-byte partial sums exchanged
    inside the library; on a windows group the rows of a Hyrax commitment are dealt out whole.  `devices` may repeat an ordinal."""

    def __init__(self, curve, bases: Buf, devices, n: Optional[int] = None, *, split: int = SPLIT_WINDOWS, exchange: int = 0,
                 window_bits: int = 0, bucket_groups: int = 1, chunk: int = 0, byte_tables: int = 0, scalars: int = SCALARS_EACH):
        self.curve = curve_id(curve)
        self._lib = _ffi.load()
        if n is None:
            if not isinstance(bases, np.ndarray):
                raise ValueError("n is required for device-resident bases")
            n = bases.shape[0]
        loc, ptr = _loc_ptr(bases, 64 * n)
        opts = MsmOpts(window_bits, bucket_groups, chunk, byte_tables, -1, (ctypes.c_uint32 * 3)(0, 0, 0))
        gopts = _ffi.GroupOpts(split, exchange, scalars, (ctypes.c_uint32 * 5)(*([0] * 5)))
        devs = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        check(self._lib.reef_msm_group_create(ctypes.byref(h), self.curve, ptr, n, loc, ctypes.byref(opts), devs, len(devices), ctypes.byref(gopts)))
        self._h = h
        self.n = n

    def enable_timing(self, on: bool = True) -> None:
        check(self._lib.reef_msm_group_enable_timing(self._h, int(on)))

    def last_timing(self) -> dict:
        """Where the last split call's time went (reef_msm_group_timing: milliseconds)."""
        t = _ffi.GroupTiming()
        check(self._lib.reef_msm_group_last_timing(self._h, ctypes.byref(t)))
        m = min(t.members, 16)
        return {"total_ms": t.total_ms, "distribute_ms": t.distribute_ms, "members_done_ms": t.members_done_ms, "combine_ms": t.combine_ms,
                "member_issue_ms": list(t.member_issue_ms[:m]), "member_stream_ms": list(t.member_stream_ms[:m])}

    def info(self) -> dict:
        gi = _ffi.GroupInfo()
        check(self._lib.reef_msm_group_info_get(self._h, ctypes.byref(gi)))
        return {"members": gi.members, "distinct_devices": gi.distinct_devices, "split": ("windows", "points")[gi.split],
                "exchange": {1: "peer", 2: "host-staged", 3: "rccl"}[gi.exchange], "peer_members": gi.peer_members,
                "key_points": list(gi.key_points[:min(gi.members, 16)])}

    def msm(self, scalars: Buf, n: Optional[int] = None, *, is_mont: bool = True) -> np.ndarray:
        if n is None:
            if not isinstance(scalars, np.ndarray):
                raise ValueError("n is required for device-resident scalars")
            n = scalars.shape[0]
        loc, ptr = _loc_ptr(scalars, 32 * n)
        out = np.zeros(12, dtype=np.uint64)
        check(self._lib.reef_msm_group_msm(self._h, ptr, n, loc, bool(is_mont), out.ctypes.data))
        return out

    def msm_rows(self, scalars: Buf, rows: int, row_len: int, *, is_mont: bool = True, max_scalar_bits: int = 0,
                 blinds: Optional[Buf] = None, h: Optional[Buf] = None) -> np.ndarray:
        loc, ptr = _loc_ptr(scalars, 32 * rows * row_len)
        bptr = hptr = None
        if blinds is not None:
            bloc, bptr = _loc_ptr(blinds, 32 * rows)
            hloc, hptr = _loc_ptr(h, 64)
            if bloc != loc or hloc != loc:
                raise ValueError("blinds and h must live where the scalars live")
        out = np.zeros((rows, 12), dtype=np.uint64)
        check(self._lib.reef_msm_group_rows(self._h, ptr, rows, row_len, loc, bool(is_mont), max_scalar_bits, bptr, hptr, out.ctypes.data))
        return out

    def msm_rows_symbols(self, symbols: Buf, rows: int, row_len: int, symbol_bits: int, *, blinds: Optional[Buf] = None,
                         h: Optional[Buf] = None, blinds_are_mont: bool = True) -> np.ndarray:
        if isinstance(symbols, np.ndarray) and symbols.dtype != np.uint8:
            raise TypeError("symbols must be uint8")
        loc, ptr = _loc_ptr(symbols, rows * row_len)
        bptr = hptr = None
        if blinds is not None:
            bloc, bptr = _loc_ptr(blinds, 32 * rows)
            hloc, hptr = _loc_ptr(h, 64)
            if bloc != loc or hloc != loc:
                raise ValueError("blinds and h must live where the symbols live")
        out = np.zeros((rows, 12), dtype=np.uint64)
        check(self._lib.reef_msm_group_rows_symbols(self._h, ptr, rows, row_len, loc, symbol_bits, bptr, hptr, bool(blinds_are_mont), out.ctypes.data))
        return out

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.reef_msm_group_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def msm_multi(ctxs, scalars, is_mont: bool = True) -> np.ndarray:
    """Several MSMs at once (reef_msm_multi): ctxs[i].msm(scalars[i]) for distinct contexts, host scalars, all enqueued before any
    is waited for.  Returns the commitments as a (count, 12) array."""
    lib = _ffi.load()
    cnt = len(ctxs)
    arrs = [np.ascontiguousarray(s, dtype=np.uint64) for s in scalars]
    hs = (ctypes.c_void_p * cnt)(*[c._h for c in ctxs])
    ps = (ctypes.c_void_p * cnt)(*[a.ctypes.data for a in arrs])
    ns = (ctypes.c_size_t * cnt)(*[a.shape[0] for a in arrs])
    out = np.zeros((cnt, 12), dtype=np.uint64)
    check(lib.reef_msm_multi(cnt, hs, ps, ns, REEF_HOST, bool(is_mont), out.ctypes.data))
    return out


def mult_pippenger(curve, points: np.ndarray, scalars: np.ndarray, is_mont: bool = True) -> np.ndarray:
    """The pasta-msm drop-in symbol (stateless; aborts the process on failure, like the
    reference panics)."""
    lib = _ffi.load()
    n = points.shape[0] if points.size else 0
    if scalars.shape[0] != n:
        raise ValueError("length mismatch")  # the Rust wrapper panics on this
    out = np.zeros(12, dtype=np.uint64)
    fn = lib.mult_pippenger_pallas if curve_id(curve) == PALLAS else lib.mult_pippenger_vesta
    pts = np.ascontiguousarray(points, dtype=np.uint64)
    sc = np.ascontiguousarray(scalars, dtype=np.uint64)
    fn(out.ctypes.data, pts.ctypes.data if n else None, n, sc.ctypes.data if n else None, bool(is_mont))
    return out


def fold(curve, gens: Buf, half: int, w1: int, w2: int, out: Optional[Buf] = None) -> Buf:
    loc, ptr = _loc_ptr(gens, 128 * half)
    if out is None:
        out = np.zeros((half, 8), dtype=np.uint64) if loc == REEF_HOST else DeviceBuffer(64 * max(half, 1))
    oloc, optr = _loc_ptr(out, 64 * half)
    if oloc != loc:
        raise ValueError("gens and out must live in the same place")
    a, b = scalar_to_limbs(w1), scalar_to_limbs(w2)
    check(_ffi.load().reef_fold(curve_id(curve), ptr, half, loc, a.ctypes.data, b.ctypes.data, optr))
    return out


def normalize(curve, jac: Buf, n: Optional[int] = None, *, affine: bool = True, compressed: bool = False):
    """Jacobian -> (affine (n,8) uint64, compressed (n,32) uint8); host in, host out."""
    if isinstance(jac, np.ndarray):
        jac = np.ascontiguousarray(jac.reshape(-1, 12))
        n = jac.shape[0]
    loc, ptr = _loc_ptr(jac, 96 * n)
    if loc != REEF_HOST:
        raise ValueError("normalize(): host buffers only in the Python front-end")
    aff = np.zeros((n, 8), dtype=np.uint64) if affine else None
    comp = np.zeros((n, 32), dtype=np.uint8) if compressed else None
    check(_ffi.load().reef_normalize(curve_id(curve), ptr, n, loc, aff.ctypes.data if affine else None,
                                     comp.ctypes.data if compressed else None))
    return aff, comp


def compress(curve, jac: np.ndarray) -> bytes:
    return normalize(curve, jac, affine=False, compressed=True)[1].tobytes()


def sum_points(curve, jac: Buf, n: Optional[int] = None, out: Optional[Buf] = None) -> Buf:
    if isinstance(jac, np.ndarray):
        jac = np.ascontiguousarray(jac.reshape(-1, 12))
        n = jac.shape[0]
    loc, ptr = _loc_ptr(jac, 96 * n)
    if out is None:
        out = np.zeros(12, dtype=np.uint64) if loc == REEF_HOST else DeviceBuffer(96)
    oloc, optr = _loc_ptr(out, 96)
    if oloc != loc:
        raise ValueError("input and out must live in the same place")
    check(_ffi.load().reef_sum_points(curve_id(curve), ptr, n, loc, optr))
    return out


def gen_bases(curve, k0: int, d: int, n: int, device: bool = False) -> Buf:
    """B_i = (k0 + i*d)*G generated on the GPU."""
    out = DeviceBuffer(64 * max(n, 1)) if device else np.zeros((n, 8), dtype=np.uint64)
    loc, ptr = _loc_ptr(out)
    check(_ffi.load().reef_gen_bases(curve_id(curve), k0, d, n, ptr, loc))
    return out


def gen_scalars(curve, seed: int, n: int, kind: int = 0, small_bound: int = 0, mont: bool = True, device: bool = False) -> Buf:
    out = DeviceBuffer(32 * max(n, 1)) if device else np.zeros((n, 4), dtype=np.uint64)
    loc, ptr = _loc_ptr(out)
    check(_ffi.load().reef_gen_scalars(curve_id(curve), seed, kind, small_bound, n, bool(mont), ptr, loc))
    return out


def plan_for(n: int, window_bits: int = 0, bucket_groups: int = 0) -> dict:
    """The plan the engine would pick (pure host logic; needs no GPU)."""
    c, w, g, t = (ctypes.c_uint32() for _ in range(4))
    check(_ffi.load().reef_msm_plan_for(n, window_bits, bucket_groups, ctypes.byref(c), ctypes.byref(w), ctypes.byref(g),
                                        ctypes.byref(t)))
    return {"window_bits": c.value, "windows": w.value, "bucket_groups": g.value, "tables": t.value}


def device_count() -> int:
    return _ffi.load().reef_device_count()


def set_device(ordinal: int) -> None:
    check(_ffi.load().reef_set_device(ordinal))


def device_sync() -> None:
    check(_ffi.load().reef_device_sync())
