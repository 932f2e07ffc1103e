"""What profiles/ says about the FINAL build must have been measured on the final build: the soak and the stress run record the
fingerprint of the library's sources (reef_amd/_ffi.py: library_sources_sha16), and this test compares it with the tree's
(round 4's soak predated three kernel edits, one of them the removal of a fence)."""
import os
import re

from reef_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _recorded(name):
    path = os.path.join(ROOT, "profiles", name)
    assert os.path.exists(path), f"profiles/{name} is missing: run tools/collect_profiles.sh on the GPU box after the last source edit"
    m = re.search(r"library sources ([0-9a-f]{16})", open(path).read())
    assert m, f"profiles/{name} does not record the library's source fingerprint"
    return m.group(1)


def test_the_committed_soak_is_of_this_build():
    assert _recorded("r06_soak.txt") == _ffi.library_sources_sha16(), "the library's sources changed after the soak: soak again (tools/soak.py) and commit it"


def test_the_committed_stress_run_is_of_this_build():
    assert _recorded("r06_sc_stress.txt") == _ffi.library_sources_sha16(), "the library's sources changed after the stress run: tools/collect_profiles.sh"
