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
    return lib
