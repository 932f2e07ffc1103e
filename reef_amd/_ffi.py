"""ctypes binding of libreef_msm.so (the C ABI declared in include/reef_msm.h).

The library is built in-tree by `build()`; nothing here falls back to a CPU implementation: if the shared object is
missing, loading raises.  Two builds of the same sources exist (csrc/Makefile): reef_amd/_lib/libreef_msm.so is the
RELEASE build -- what load() binds, what bench.py measures, what the GPU suite tests -- and libreef_msm_exp.so carries
the A/B switches of common.h (load_experiment(): the few tests that force a code path, and tools/).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_bool, c_char_p, c_double, c_float, c_int, c_int32, c_size_t, c_uint8, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# the library asks the HIP runtime for eight hardware queues when it is loaded (api.cpp); a process that initialises HIP before
# loading it (torch) gets the same if this module is imported first
if os.environ.get("REEF_MSM_HW_QUEUES", "") != "0":
    os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("REEF_MSM_HW_QUEUES") or "8")
LIB_PATH = os.environ.get("REEF_MSM_LIB") or os.path.join(_HERE, "_lib", "libreef_msm.so")   # REEF_MSM_LIB: another build of the same library (tools/: the experiment build)
EXPERIMENT_LIB_PATH = os.path.join(_HERE, "_lib", "libreef_msm_exp.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "reef_msm.h")

REEF_HOST, REEF_DEVICE = 0, 1
PALLAS, VESTA = 0, 1
STATUS_NAMES = {0: "REEF_OK", 1: "REEF_ERR_ARG", 2: "REEF_ERR_HIP", 3: "REEF_ERR_NO_GPU", 4: "REEF_ERR_OOM"}


class ReefError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class MsmOpts(ctypes.Structure):
    _fields_ = [("window_bits", c_uint32), ("bucket_groups", c_uint32), ("chunk", c_uint32), ("byte_tables", c_uint32),
                ("device", c_int32), ("reserved", c_uint32 * 3)]


class RuntimeOpts(ctypes.Structure):
    _fields_ = [("hw_queues", c_int32), ("warm", c_uint32), ("reserved", c_uint32 * 6)]


class RuntimeInfo(ctypes.Structure):
    _fields_ = [("hw_queues_env", c_int32), ("hw_queues_set_by_library", c_int32), ("abi_version", c_uint32), ("warm", c_uint32)]


class GroupOpts(ctypes.Structure):
    _fields_ = [("split", c_uint32), ("exchange", c_uint32), ("scalars", c_uint32), ("reserved", c_uint32 * 5)]


class GroupInfo(ctypes.Structure):
    _fields_ = [("members", c_uint32), ("distinct_devices", c_uint32), ("split", c_uint32), ("exchange", c_uint32), ("peer_members", c_uint32),
                ("reserved", c_uint32 * 3), ("key_points", c_uint64 * 16)]


class GroupTiming(ctypes.Structure):
    _fields_ = [("members", c_uint32), ("reserved", c_uint32), ("total_ms", c_double), ("distribute_ms", c_double), ("members_done_ms", c_double),
                ("combine_ms", c_double), ("member_issue_ms", c_double * 16), ("member_stream_ms", c_double * 16)]


class KeyCacheTiming(ctypes.Structure):
    _fields_ = [(n, c_uint64) for n in ("calls", "nominate_ns", "enqueue_ns", "confirm_ns", "wait_ns")] + [("reserved", c_uint64 * 3)]


class KeyCacheStats(ctypes.Structure):
    _fields_ = [(n, c_uint64) for n in ("entries", "resident_keys", "resident_bytes", "builds", "hits", "clones", "misspeculated", "spares")]


ABI_VERSION = 6     # REEF_ABI_VERSION of include/reef_msm.h this binding was written against


def build(force: bool = False, jobs: int = 4) -> str:
    """Compile every HIP source for gfx950, both builds (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, f"-j{jobs}"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for path in (os.path.join(_HERE, "_lib", "libreef_msm.so"), EXPERIMENT_LIB_PATH):
        if not os.path.exists(path):
            raise RuntimeError("build finished but %s is missing" % path)
    return LIB_PATH


_lib = None
_exp_lib = None


def _bind(path: str) -> ctypes.CDLL:
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the MSM path)")
    lib = ctypes.CDLL(path)
    vp = c_void_p
    sig = {
        "mult_pippenger_pallas": (None, [vp, vp, c_size_t, vp, c_bool]),
        "mult_pippenger_vesta": (None, [vp, vp, c_size_t, vp, c_bool]),
        "reef_msm_ctx_create": (c_int, [POINTER(vp), c_int, vp, c_size_t, c_int, POINTER(MsmOpts)]),
        "reef_msm_ctx_clone": (c_int, [POINTER(vp), vp]),
        "reef_msm_ctx_attach": (c_int, [vp, vp]),
        "reef_msm_ctx_set_bases": (c_int, [vp, vp, c_size_t, c_int]),
        "reef_msm_ctx_destroy": (None, [vp]),
        "reef_msm_ctx_sync": (c_int, [vp]),
        "reef_msm_ctx_stream": (vp, [vp]),
        "reef_msm": (c_int, [vp, vp, c_size_t, c_int, c_bool, vp, c_int]),
        "reef_msm_multi": (c_int, [c_size_t, POINTER(vp), POINTER(vp), POINTER(c_size_t), c_int, c_bool, vp]),
        "reef_msm_rows": (c_int, [vp, vp, c_size_t, c_size_t, c_int, c_bool, c_uint32, vp, vp, vp, c_int]),
        "reef_msm_rows_symbols": (c_int, [vp, vp, c_size_t, c_size_t, c_int, c_uint32, vp, vp, c_bool, vp, c_int]),
        "reef_ipa_cross_terms": (c_int, [vp, vp, c_size_t, c_int, c_bool, vp, vp, c_size_t, vp, vp]),
        "reef_fold": (c_int, [c_int, vp, c_size_t, c_int, vp, vp, vp]),
        "reef_normalize": (c_int, [c_int, vp, c_size_t, c_int, vp, vp]),
        "reef_mle_bound_rows": (c_int, [c_int, vp, c_size_t, c_int, c_int, c_bool, vp, c_size_t, c_size_t, vp, c_int, vp]),
        "reef_sum_points": (c_int, [c_int, vp, c_size_t, c_int, vp]),
        "reef_gen_bases": (c_int, [c_int, c_uint64, c_uint64, c_size_t, vp, c_int]),
        "reef_gen_scalars": (c_int, [c_int, c_uint64, c_int, c_uint64, c_size_t, c_bool, vp, c_int]),
        "reef_device_count": (c_int, []),
        "reef_set_device": (c_int, [c_int]),
        "reef_get_device": (c_int, [POINTER(c_int)]),
        "reef_device_sync": (c_int, []),
        "reef_device_alloc": (vp, [c_size_t]),
        "reef_device_free": (None, [vp]),
        "reef_memcpy": (c_int, [vp, vp, c_size_t, c_int, c_int]),
        "reef_last_error": (c_char_p, []),
        "reef_version": (c_char_p, []),
        "reef_abi_version": (c_uint32, []),
        "reef_key_cache_info": (None, [POINTER(KeyCacheStats)]),
        "reef_key_cache_clear": (None, []),
        "reef_key_cache_wait": (None, []),
        "reef_key_cache_timing_get": (None, [POINTER(KeyCacheTiming), c_int]),
        "reef_runtime_init": (c_int, [POINTER(RuntimeOpts), POINTER(RuntimeInfo)]),
        "reef_msm_ctx_last_timing": (c_int, [vp, POINTER(c_float), POINTER(c_float)]),
        "reef_msm_ctx_enable_timing": (c_int, [vp, c_int]),
        "reef_msm_ctx_set_window_split": (c_int, [vp, c_uint32, c_uint32]),
        "reef_msm_ctx_timing_stats": (c_int, [vp, c_int, POINTER(c_uint64), POINTER(c_double), POINTER(c_double)]),
        "reef_msm_ctx_sum_points": (c_int, [vp, vp, c_size_t, vp]),
        "reef_msm_ctx_plan": (c_int, [vp, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32)]),
        "reef_msm_ctx_byte_tables": (c_int, [vp]),
        "reef_msm_plan_for": (c_int, [c_size_t, c_uint32, c_uint32, POINTER(c_uint32), POINTER(c_uint32), POINTER(c_uint32),
                                      POINTER(c_uint32)]),
        "reef_msm_group_create": (c_int, [POINTER(vp), c_int, vp, c_size_t, c_int, POINTER(MsmOpts), POINTER(c_int), c_size_t, POINTER(GroupOpts)]),
        "reef_msm_group_destroy": (None, [vp]),
        "reef_msm_group_info_get": (c_int, [vp, POINTER(GroupInfo)]),
        "reef_msm_group_enable_timing": (c_int, [vp, c_int]),
        "reef_msm_group_last_timing": (c_int, [vp, POINTER(GroupTiming)]),
        "reef_msm_group_msm": (c_int, [vp, vp, c_size_t, c_int, c_bool, vp]),
        "reef_msm_group_rows": (c_int, [vp, vp, c_size_t, c_size_t, c_int, c_bool, c_uint32, vp, vp, vp]),
        "reef_msm_group_rows_symbols": (c_int, [vp, vp, c_size_t, c_size_t, c_int, c_uint32, vp, vp, c_bool, vp]),
        "reef_msm_folded": (c_int, [vp, vp, c_size_t, c_size_t, c_int, c_bool, vp, vp, c_size_t, vp, c_int]),
        "reef_sc_create": (c_int, [POINTER(vp), c_int, c_size_t]),
        "reef_sc_destroy": (None, [vp]),
        "reef_sc_set_table": (c_int, [vp, c_int, vp, c_size_t, c_int]),
        "reef_sc_gen_eq_table": (c_int, [vp, vp, vp, c_size_t, vp, c_size_t]),
        "reef_sc_round_coeffs": (c_int, [vp, c_size_t, vp]),
        "reef_sc_fold": (c_int, [vp, c_size_t, vp]),
        "reef_sc_fold_and_next_coeffs": (c_int, [vp, c_size_t, vp, vp]),
        "reef_sc_read": (c_int, [vp, c_int, c_size_t, vp]),
        "reef_sc_reset_table": (c_int, [vp]),
        "reef_sc_sync": (c_int, [vp]),
        "reef_merkle_nodes": (c_uint64, [c_uint64]),
        "reef_merkle_commit": (c_int, [c_int, vp, vp, c_size_t, c_int, c_bool, vp, c_int, vp]),
        "reef_merkle_commit_devices": (c_int, [c_int, vp, vp, c_size_t, c_bool, vp, c_size_t, vp, vp, vp]),
        "reef_derive_generators": (c_int, [c_int, vp, c_size_t, c_size_t, vp, c_bool, vp, c_int]),
        "reef_shake256": (None, [vp, c_size_t, vp, c_size_t]),
        "reef_test_field_op": (c_int, [c_int, c_int, vp, vp, vp, c_size_t]),
        "reef_test_ec_op": (c_int, [c_int, c_int, vp, vp, vp, vp, c_size_t]),
        "reef_bench_fmul": (c_int, [c_int, c_uint32, POINTER(c_double)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.reef_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path} has ABI version {lib.reef_abi_version()}, this binding expects {ABI_VERSION}: rebuild")
    return lib


def load() -> ctypes.CDLL:
    """The product library (the release build unless REEF_MSM_LIB names another)."""
    global _lib
    if _lib is None:
        _lib = _bind(LIB_PATH)
    return _lib


def load_experiment() -> ctypes.CDLL:
    """The +experiment build of the same sources, beside the product library in the same process (its own streams and caches)."""
    global _exp_lib
    if _exp_lib is None:
        _exp_lib = _bind(EXPERIMENT_LIB_PATH)
    return _exp_lib


def is_release(lib=None) -> bool:
    return b"+experiment" not in (lib or load()).reef_version()


def check(status: int) -> None:
    if status != 0:
        raise ReefError(status, load().reef_last_error().decode(errors="replace"))


def kernel_sources_sha16() -> str:
    """Fingerprint of the sources the MSM kernels are compiled from: a PMC profile in profiles/ describes THESE kernels or it is
    stale (bench.py refuses a stale one for roofline.traffic; tools/pmc_traffic.py records the fingerprint it ran on)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("msm_kernels.inc", "engine.inc", "field.h", "ec.h", "ec_coop.h", "field_mad_gfx950.h", "Makefile"):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def library_sources_sha16() -> str:
    """Fingerprint of EVERYTHING libreef_msm.so is compiled from (kernels, engines, the C ABI, the Makefile): what a soak or a stress run
    vouches for is this build and no other (profiles/r05_soak.txt records it; tests/test_profiles_fresh.py compares)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    names = sorted(glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")) +
                   glob.glob(os.path.join(CSRC, "*.cpp")) + [os.path.join(CSRC, "Makefile"), HEADER])
    for name in names:
        with open(name, "rb") as f:
            h.update(os.path.basename(name).encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def declared_symbols() -> list[str]:
    """Function names declared in include/reef_msm.h (used by the ABI export test)."""
    import re
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b((?:reef_|mult_pippenger_)\w+)\s*\(", text)))
