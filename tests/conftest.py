"""Test configuration.

`-m "not gpu"`: oracle vs golden vectors, host logic, ABI export check (runs anywhere).
`-m gpu`     : parity of the HIP path (through the C ABI) against the oracle / golden vectors.
"""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_sessionstart(session):
    """Fresh checkout: build the product library (hipcc cross-compiles without a GPU) and the
    oracle's C restatement once; both are git-ignored artefacts."""
    from reef_amd import _ffi
    if not os.path.exists(_ffi.LIB_PATH) and os.path.exists("/opt/rocm/bin/hipcc"):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "pasta_msm_golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def cref():
    """The C restatement (oracle/libpasta_ref.so), built on demand."""
    from oracle import pasta_ref
    pasta_ref.lib()
    return pasta_ref


@pytest.fixture(scope="session")
def gpu_lib():
    """The product library; GPU tests must fail (not skip) if it is missing."""
    from reef_amd import _ffi
    lib = _ffi.load()
    assert lib.reef_device_count() > 0, "no HIP device visible"
    assert _ffi.is_release(lib) or os.environ.get("REEF_MSM_LIB"), f"the GPU suite tests the release build, not {lib.reef_version()!r}"
    return lib


@pytest.fixture
def experiment_build(monkeypatch):
    """Opt-in for the few tests that FORCE a code path with an A/B switch of common.h (exp_env): for the test's duration the Python front-ends
    bind libreef_msm_exp.so (-DREEF_EXPERIMENT).  Every other GPU test -- and bench.py -- runs on the release build, where those switches do
    not exist."""
    from reef_amd import _ffi
    exp = _ffi.load_experiment()
    assert not _ffi.is_release(exp), "libreef_msm_exp.so was built without -DREEF_EXPERIMENT"
    monkeypatch.setattr(_ffi, "_lib", exp)
    return exp


@pytest.fixture(params=["release", "experiment"])
def either_build(request, monkeypatch):
    """Tests that set experiment switches but whose assertions are plain parity with the oracle run on BOTH builds: on the release build the
    switches do not exist, so the same inputs go down the shipped code path."""
    from reef_amd import _ffi
    if request.param == "experiment":
        monkeypatch.setattr(_ffi, "_lib", _ffi.load_experiment())
    return request.param
