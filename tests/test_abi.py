"""The C-ABI library loads and exports every symbol include/reef_msm.h declares; pure host
logic reachable without a GPU behaves.  No compute calls here."""
import ctypes
import os

import pytest

from reef_amd import _ffi, msm


def test_library_present_and_loads():
    assert os.path.exists(_ffi.LIB_PATH), "run __graft_entry__.build() first"
    lib = _ffi.load()
    assert lib.reef_version().startswith(b"reef_msm")
    import re
    header = int(re.search(r"#define REEF_ABI_VERSION (\d+)", open(_ffi.HEADER).read()).group(1))
    assert lib.reef_abi_version() == header == _ffi.ABI_VERSION          # header, binding and binary agree


def test_the_loaded_build_is_the_release_build_and_says_so():
    """The A/B switches (common.h: exp_env) exist only in builds with -DREEF_EXPERIMENT.  Since round 6 the library the binding loads -- the one
    bench.py measures and the GPU suite tests -- is the RELEASE build (reef_amd/_lib/libreef_msm.so); the switches live in libreef_msm_exp.so,
    bound only by load_experiment().  The version string tells the two apart."""
    if not os.environ.get("REEF_MSM_LIB"):
        assert _ffi.is_release() and b"release" in _ffi.load().reef_version()
    assert b"+experiment" in _ffi.load_experiment().reef_version()
    assert _ffi.load_experiment().reef_abi_version() == _ffi.load().reef_abi_version()
    mk = open(os.path.join(_ffi.CSRC, "Makefile")).read()
    assert "libreef_msm_exp.so" in mk and "release:" in mk and "-DREEF_EXPERIMENT" in mk
    common = open(os.path.join(_ffi.CSRC, "common.h")).read()
    import glob
    import re
    supported = set(re.findall(r"REEF_[A-Z0-9_]+", common.split("inline const char *exp_env")[0].split("Environment switches come in two kinds")[1]))
    supported -= {"REEF_EXPERIMENT"}
    read = set()
    for f in glob.glob(os.path.join(_ffi.CSRC, "*.inc")) + glob.glob(os.path.join(_ffi.CSRC, "*.cpp")) + glob.glob(os.path.join(_ffi.CSRC, "*.h")):
        read |= set(re.findall(r'[^_a-z]getenv\("(REEF_[A-Z0-9_]+)"\)', open(f).read()))
    assert read <= supported | {"REEF_MSM_KEY_CACHE_MB", "REEF_MSM_KEY_HOST_MB"}, read - supported      # every plain getenv is a documented switch
    doc = open(os.path.join(os.path.dirname(_ffi.HEADER), "..", "INTEGRATION.md")).read()
    assert all(name in doc for name in supported), [n for n in supported if n not in doc]


def test_runtime_init_reports_who_set_the_hardware_queues():
    """reef_runtime_init makes the GPU_MAX_HW_QUEUES choice explicit: it never overwrites a value the user exported and
    reports what the environment holds; pure host logic, no HIP call."""
    lib = _ffi.load()
    info = _ffi.RuntimeInfo()
    assert lib.reef_runtime_init(None, ctypes.byref(info)) == 0
    assert info.abi_version == _ffi.ABI_VERSION
    assert info.hw_queues_env == int(os.environ.get("GPU_MAX_HW_QUEUES", "0"))   # reef_amd._ffi (or the user) exported it before the load
    assert info.hw_queues_set_by_library == 0                                  # so the library's constructor left it alone
    assert info.warm == 0                                                        # nobody asked for the warm-up (REEF_MSM_WARM unset)
    bad = _ffi.RuntimeOpts(hw_queues=0, warm=3)
    assert lib.reef_runtime_init(ctypes.byref(bad), None) == 1 and b"warm" in lib.reef_last_error()
    opts = _ffi.RuntimeOpts(hw_queues=16)
    assert lib.reef_runtime_init(ctypes.byref(opts), ctypes.byref(info)) == 0
    assert info.hw_queues_env == int(os.environ.get("GPU_MAX_HW_QUEUES", "0")) and info.hw_queues_set_by_library == 0


def test_exports_every_declared_symbol():
    names = _ffi.declared_symbols()
    assert "mult_pippenger_pallas" in names and "reef_msm_rows" in names and len(names) >= 25
    raw = ctypes.CDLL(_ffi.LIB_PATH)
    missing = [n for n in names if not hasattr(raw, n)]
    assert not missing, missing


def test_struct_sizes_match_header():
    assert ctypes.sizeof(_ffi.MsmOpts) == 32
    assert ctypes.sizeof(_ffi.RuntimeOpts) == 32 and ctypes.sizeof(_ffi.RuntimeInfo) == 16


def test_plan_for_host_logic():
    # pure host planning logic: W = ceil(256/c), T = ceil(W/G)
    p = msm.plan_for(1 << 20)
    assert p["windows"] == -(-256 // p["window_bits"]) and p["bucket_groups"] == p["windows"] and p["tables"] == 1
    assert 12 <= p["window_bits"] <= 18
    p = msm.plan_for(1 << 20, bucket_groups=1)
    assert p["bucket_groups"] == 1 and p["tables"] == p["windows"]
    p = msm.plan_for(1 << 16, window_bits=13, bucket_groups=4)
    assert p == {"window_bits": 13, "windows": 20, "bucket_groups": 4, "tables": 5}
    small, big = msm.plan_for(200)["window_bits"], msm.plan_for(1 << 22)["window_bits"]
    assert small < big
    with pytest.raises(msm.ReefError):
        msm.plan_for(100, window_bits=25)


def test_every_default_plan_is_runnable_up_to_the_key_limit():
    """A plan the engine hands out must be one its pipeline accepts: the bucket keys of one MSM fit the scan
    (G * 2^(c-1) <= 4 Mi keys) and its digits a 32-bit index, for every key length reef_msm_ctx_create
    accepts (n < 2^27) and every bucket-group mode; what cannot run is rejected when the key is made."""
    max_keys = 4 * 1024 * 1024
    for logn in range(0, 27):
        for n in {1 << logn, (1 << logn) + 1, (1 << (logn + 1)) - 1}:
            if n >= 1 << 27:
                continue
            for groups in (0, 1, 2, 4, 7):
                p = msm.plan_for(n, bucket_groups=groups)
                c, w, g, t = p["window_bits"], p["windows"], p["bucket_groups"], p["tables"]
                assert 2 <= c <= 20 and w == -(-256 // c) and t == -(-w // g)
                assert (g << (c - 1)) <= max_keys, (n, groups, p)
                assert n * w < 1 << 32, (n, groups, p)
    assert msm.plan_for((1 << 27) - 1)["window_bits"] == 19      # the cost model asks for 20 there; 13 * 2^19 keys do not fit
    with pytest.raises(msm.ReefError):
        msm.plan_for(1 << 27)                                     # beyond the key limit
    with pytest.raises(msm.ReefError):
        msm.plan_for(1 << 24, window_bits=20)                     # an explicit plan that cannot run is refused, not clamped
    assert msm.plan_for(1 << 24, window_bits=20, bucket_groups=1)["tables"] == 13


def test_struct_sizes_of_the_group_api_and_its_argument_errors():
    """reef_msm_group_create (include/reef_msm.h section 5) rejects bad arguments before it touches a device: pure host logic."""
    import numpy as np
    assert ctypes.sizeof(_ffi.GroupOpts) == 32 and ctypes.sizeof(_ffi.GroupInfo) == 32 + 128
    lib = _ffi.load()
    bases = np.zeros((4, 8), dtype=np.uint64)
    h = ctypes.c_void_p()
    devs = (ctypes.c_int * 2)(0, 0)

    def create(curve=0, ptr=bases.ctypes.data, n=4, d=devs, nd=2, g=None):
        return lib.reef_msm_group_create(ctypes.byref(h), curve, ptr, n, 0, None, d, nd, g)
    assert create(curve=7) == 1 and b"curve" in lib.reef_last_error()
    assert create(ptr=None) == 1
    assert create(d=None) == 1
    assert create(nd=0) == 1 and create(nd=65) == 1 and b"members" in lib.reef_last_error()
    assert create(g=ctypes.byref(_ffi.GroupOpts(2, 0))) == 1 and b"split" in lib.reef_last_error()
    assert create(g=ctypes.byref(_ffi.GroupOpts(0, 4))) == 1 and b"exchange" in lib.reef_last_error()
    assert create(g=ctypes.byref(_ffi.GroupOpts(0, 0, 2))) == 1 and b"scalars" in lib.reef_last_error()          # REEF_SCALARS_*: 0 or 1
    assert ctypes.sizeof(_ffi.GroupTiming) == 8 + 4 * 8 + 2 * 16 * 8
    assert lib.reef_msm_group_enable_timing(None, 1) == 1 and lib.reef_msm_group_last_timing(None, None) == 1
    assert create(d=(ctypes.c_int * 2)(0, -1)) == 1
    assert lib.reef_msm_group_create(None, 0, bases.ctypes.data, 4, 0, None, devs, 2, None) == 1
    if lib.reef_device_count() == 0:
        assert create() == 3 and h.value is None                    # REEF_ERR_NO_GPU: fails loudly, nothing is created
    assert lib.reef_msm_group_msm(None, None, 0, 0, True, None) == 1
    assert lib.reef_msm_group_info_get(None, None) == 1
    lib.reef_msm_group_destroy(None)                                # a no-op
    # the Merkle tree in blocks over several devices: argument errors before any device is touched
    doc = np.arange(8, dtype=np.uint32)
    root = np.zeros(4, dtype=np.uint64)
    assert lib.reef_merkle_commit_devices(0, None, doc.ctypes.data, 8, False, devs, 2, None, root.ctypes.data, None) == 1
    assert lib.reef_merkle_commit_devices(0, None, doc.ctypes.data, 8, False, None, 2, None, root.ctypes.data, None) == 1
    assert lib.reef_merkle_commit_devices(0, None, doc.ctypes.data, 8, False, devs, 0, None, root.ctypes.data, None) == 1
    assert lib.reef_merkle_commit_devices(9, None, doc.ctypes.data, 8, False, devs, 2, None, root.ctypes.data, None) == 1


def test_unknown_curve_rejected():
    with pytest.raises(ValueError):
        msm.curve_id("bls12")


def test_fails_loudly_without_gpu():
    lib = _ffi.load()
    if lib.reef_device_count() > 0:
        pytest.skip("a GPU is present")
    import numpy as np
    with pytest.raises(msm.ReefError) as e:
        msm.MsmContext("pallas", np.zeros((4, 8), dtype=np.uint64))
    assert e.value.status == 3  # REEF_ERR_NO_GPU: no silent CPU fallback


def test_replay_library_exports_its_entry_point_and_fails_loudly_without_gpu():
    """libreef_replay.so (the C++ host that issues Reef's MSM sequence) exports reef_replay_run; without a GPU it returns an
    error code and a message, never a JSON line; a missing shapes file is an error as well."""
    from reef_amd import replay
    assert os.path.exists(replay.LIB_PATH), "run __graft_entry__.build() first"
    assert os.path.exists(replay.SHAPES_PATH)
    lib = ctypes.CDLL(replay.LIB_PATH)
    assert hasattr(lib, "reef_replay_run")
    if _ffi.load().reef_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError) as e:
        replay.run("cfg3")
    assert "failed with 3" in str(e.value)
    shapes = replay.shapes()
    assert set(n.split("_")[0] for n in shapes) == {"cfg1", "cfg3", "cfg4", "cfg4b", "cfg5"}


def test_warm_up_without_a_gpu_fails_quietly_and_says_so():
    """REEF_MSM_WARM=1 / reef_runtime_init({warm}) on a box without a usable device: the warm-up thread ends with state 3 (failed), nothing aborts, and the
    first real call is the one that reports why (no CPU fallback).  A fresh interpreter: the request is read when the library is loaded."""
    import subprocess
    import sys
    code = ("import ctypes, time\n"
            "from reef_amd import _ffi\n"
            "lib = _ffi.load(); info = _ffi.RuntimeInfo()\n"
            "for _ in range(100):\n"
            "    lib.reef_runtime_init(None, ctypes.byref(info))\n"
            "    if info.warm != 1: break\n"
            "    time.sleep(0.1)\n"
            "print('warm', info.warm, 'devices', lib.reef_device_count())\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, REEF_MSM_WARM="1"),
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr[-1500:]
    words = out.stdout.split()
    state, devices = int(words[1]), int(words[3])
    assert state == (2 if devices > 0 else 3), out.stdout
