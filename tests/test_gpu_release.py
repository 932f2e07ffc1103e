"""The RELEASE build (`make -C reef_amd/csrc release` -> reef_amd/_lib/libreef_msm_release.so: the library an embedder ships, without
-DREEF_EXPERIMENT, so none of the A/B switches of common.h exist in it) computes what the in-tree experiment build computes.  The rest
of the GPU suite loads the experiment build (its tests drive those switches); here a fresh interpreter loads the release build through
REEF_MSM_LIB and runs (1) __graft_entry__.smoke() -- every row of the path against the oracle -- and (2) MSMs at Reef's sizes and at the
bench size against their discrete logarithms, plus a switch that must NOT be read.  Skipped when the release build is absent or older
than the sources (build(): the driver's build check compiles the experiment build only; `make release` adds three minutes)."""
import glob
import os
import subprocess
import sys

import pytest

from reef_amd import _ffi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RELEASE = os.path.join(os.path.dirname(_ffi.LIB_PATH), "libreef_msm_release.so")

CHILD = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["REEF_ROOT"])
import __graft_entry__ as g
from reef_amd import _ffi, msm
lib = _ffi.load()
ver = lib.reef_version().decode()
assert _ffi.LIB_PATH.endswith("libreef_msm_release.so") and "+experiment" not in ver, (ver, _ffi.LIB_PATH)
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
print("release ok:", ver)
"""


def _fresh():
    if not os.path.exists(RELEASE):
        return False
    srcs = [f for pat in ("*.inc", "*.h", "*.hip", "*.cpp", "Makefile") for f in glob.glob(os.path.join(_ffi.CSRC, pat))] + [_ffi.HEADER]
    return os.path.getmtime(RELEASE) >= max(os.path.getmtime(f) for f in srcs)


@pytest.mark.skipif(not _fresh(), reason="reef_amd/_lib/libreef_msm_release.so absent or older than the sources: make -C reef_amd/csrc release")
def test_release_build_matches_the_oracle(gpu_lib):
    # REEF_MSM_WIDE=1 is a supported switch (byte tables by policy); REEF_MSM_SORT_COMPACT is an experiment switch the release build must not know
    env = dict(os.environ, REEF_MSM_LIB=RELEASE, REEF_ROOT=ROOT, REEF_MSM_SORT_COMPACT="0")
    out = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "smoke ok" in out.stdout and "release ok: reef_msm" in out.stdout
    raw = open(RELEASE, "rb").read()
    assert b"REEF_MSM_SORT_COMPACT" not in raw and b"REEF_SC_BLOCKS" not in raw        # the experiment switches' names are not even in the binary
    assert b"REEF_MSM_KEY_CACHE" in raw                                               # the supported ones are
