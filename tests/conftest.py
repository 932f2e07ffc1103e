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
