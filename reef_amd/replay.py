"""ctypes front-end of libreef_replay.so: the MSM sequence of one `reef --prove` run (eniac/Reef
src/backend/framework.rs:642-754) issued through the C ABI by the C++ harness reef_amd/csrc/host/reef_replay.cpp.

The harness is host code of the product side (it links libreef_msm.so only); its MSM lengths come from
tests/golden/replay_shapes.json, which oracle/gen_replay_shapes.py derives from Reef's cost model.  Every per-step
commitment is checked inside the harness against its discrete-logarithm closed form; a mismatch is an error here.
"""
from __future__ import annotations

import ctypes
import json
import os

from . import _ffi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libreef_replay.so")
SHAPES_PATH = os.path.join(os.path.dirname(_HERE), "tests", "golden", "replay_shapes.json")

_lib = None


def _load() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _ffi.load()                       # libreef_msm.so first (the replay library resolves it through its rpath as well)
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        lib = ctypes.CDLL(LIB_PATH)
        lib.reef_replay_run.restype = ctypes.c_int
        lib.reef_replay_run.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
        lib.reef_replay_run_devices.restype = ctypes.c_int
        lib.reef_replay_run_devices.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.c_size_t,
                                                ctypes.c_char_p, ctypes.c_size_t]
        _lib = lib
    return _lib


def run(config: str = "cfg3", nofold: bool = True, tables: bool = False, shapes_path: str | None = None, devices=None) -> dict:
    """One replay; returns the harness's JSON line as a dict.  Raises on any failed call or mismatching commitment.
    devices: ordinals (may repeat) for the multi-device leg -- the final SNARK's arguments placed whole on per-device contexts and the
    document commitment through a device group (reef_msm_group_*), all from this one process; line["devices"] reports it."""
    buf = ctypes.create_string_buffer(32768)
    devs = list(devices or [])
    arr = (ctypes.c_int * max(len(devs), 1))(*devs)
    rc = _load().reef_replay_run_devices((shapes_path or SHAPES_PATH).encode(), config.encode(), int(nofold), int(tables), arr, len(devs), buf, len(buf))
    text = buf.value.decode(errors="replace")
    if rc != 0:
        raise RuntimeError(f"reef_replay_run({config}) failed with {rc}: {text}")
    return json.loads(text)


def shapes(shapes_path: str | None = None) -> dict:
    with open(shapes_path or SHAPES_PATH) as f:
        return {s["name"]: s for s in json.load(f)["shapes"]}
