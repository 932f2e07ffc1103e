"""Two builds of the same sources (reef_amd/csrc/Makefile).  reef_amd/_lib/libreef_msm.so is the RELEASE build -- no -DREEF_EXPERIMENT, so none of
the A/B switches of common.h exist in it: the library an embedder ships, and since round 6 the one reef_amd/_ffi.py loads, bench.py measures and
this whole GPU suite tests.  libreef_msm_exp.so carries the switches for the few tests that force a code path (conftest.py: experiment_build).
Here: (1) the suite's library really is the release build and does not even contain the switches' names; (2) a fresh interpreter loads the
EXPERIMENT build through REEF_MSM_LIB and runs __graft_entry__.smoke() -- every row of the path against the oracle -- and MSMs at Reef's sizes and
at the bench size against their discrete logarithms: with no switch set it computes what the release build computes."""
import glob
import os
import subprocess
import sys

import pytest

from reef_amd import _ffi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RELEASE = os.path.join(os.path.dirname(_ffi.EXPERIMENT_LIB_PATH), "libreef_msm.so")
EXPERIMENT = _ffi.EXPERIMENT_LIB_PATH

CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["REEF_ROOT"])
import __graft_entry__ as g
from reef_amd import _ffi, msm
lib = _ffi.load()
ver = lib.reef_version().decode()
assert _ffi.LIB_PATH.endswith("libreef_msm_exp.so") and "+experiment" in ver, (ver, _ffi.LIB_PATH)
g.smoke()
from oracle.pasta_oracle import CURVES
sys.path.insert(0, os.environ["REEF_ROOT"])
import bench
for curve, logn in (("pallas", 15), ("vesta", 14), ("pallas", 20)):
    n = (1 << logn) - 37                       # ragged: not a power of two
    k0, d = 0xABCDEF, 0x12345
    bases = msm.gen_bases(curve, k0, d, n, device=True)
    sc = msm.gen_scalars(curve, 0x5EED, n, kind=0, mont=True, device=True)
    canon = msm.gen_scalars(curve, 0x5EED, n, kind=0, mont=False)
    want = bench.point_of_dlog(curve, bench.dlog_of_msm(curve, canon, k0, d, 0))
    for groups in (0, 1):
        with msm.MsmContext(curve, bases, n, bucket_groups=groups) as ctx:
            out = np.zeros(12, dtype=np.uint64)
            ctx.msm(sc, n, out=out)
            assert msm.compress(curve, out) == want, (curve, logn, groups)
print("experiment build ok:", ver)
"""


def _fresh(path):
    if not os.path.exists(path):
        return False
    srcs = [f for pat in ("*.inc", "*.h", "*.hip", "*.cpp", "Makefile") for f in glob.glob(os.path.join(_ffi.CSRC, pat))] + [_ffi.HEADER]
    return os.path.getmtime(path) >= max(os.path.getmtime(f) for f in srcs)


def test_the_suite_runs_on_the_release_build(gpu_lib):
    ver = gpu_lib.reef_version().decode()
    assert "release" in ver and "+experiment" not in ver, ver
    assert os.path.samefile(_ffi.LIB_PATH, RELEASE) or os.environ.get("REEF_MSM_LIB")
    assert _fresh(RELEASE), "reef_amd/_lib/libreef_msm.so is older than the sources: python -c 'import __graft_entry__ as g; g.build()'"
    raw = open(RELEASE, "rb").read()
    assert b"REEF_MSM_SORT_COMPACT" not in raw and b"REEF_SC_BLOCKS" not in raw        # the experiment switches' names are not even in the binary
    assert b"REEF_MSM_KEY_CACHE" in raw                                               # the supported ones are
    assert b"REEF_MSM_SORT_COMPACT" in open(EXPERIMENT, "rb").read()


def test_experiment_build_matches_the_oracle(gpu_lib):
    assert _fresh(EXPERIMENT), "reef_amd/_lib/libreef_msm_exp.so absent or older than the sources"
    env = dict(os.environ, REEF_MSM_LIB=EXPERIMENT, REEF_ROOT=ROOT)
    out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "smoke ok" in out.stdout and "experiment build ok: reef_msm" in out.stdout
