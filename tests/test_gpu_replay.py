"""The MSM sequence of `reef --prove` (eniac/Reef src/backend/framework.rs:664-723: prove_step per batch, CompressedSNARK::prove,
the consistency proof) replayed through the C ABI by the C++ harness, under pytest: every per-step commitment is checked
inside the harness against its discrete-logarithm closed form (host big-integer arithmetic + a one-point key), with the
per-step scalars in host memory and the commitments returned to the host, as nova hands them over.

The MSM lengths are those of tests/golden/replay_shapes.json (oracle/gen_replay_shapes.py: Reef's cost model restated,
tests/test_costs_oracle.py).  Both serving paths of a resident key are covered: the bucket pipeline and the byte tables.
"""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _expected_checks(shape):
    timed_steps = max(shape["steps"], 3)            # the harness times at least three steps
    return 2 + 4 * timed_steps + 4 + 4 + 2          # warm-up W1/W2, four commitments per step, the two batched pairs, the two concurrent pairs, two of the four at once


@pytest.mark.parametrize("cfg", ["cfg3", "cfg4"])
@pytest.mark.parametrize("tables", [False, True])
def test_replay_checks_every_commitment(cfg, tables, gpu_lib):
    from reef_amd import replay
    line = replay.run(cfg, nofold=True, tables=tables)
    shape = next(s for name, s in replay.shapes().items() if cfg in name)
    assert line["replay"] == shape["name"]
    assert (line["w1"], line["c1"], line["w2"], line["c2"]) == (shape["w1"], shape["c1"], shape["w2"], shape["c2"])   # nothing typed into the harness
    assert line["commitments_checked_against_dlog"] == _expected_checks(shape)
    assert line["steps"] == shape["steps"] and line["ipa_pallas_rounds"] == (line["key_pallas"]).bit_length() - 1
    assert line["total_prove_msm_ms"] > 0 and line["ms_per_step"] > 0
    assert ("built with the keys" in line["byte_tables"]) == tables


def test_replay_cfg5_merkle_document(gpu_lib):
    """BASELINE configs[4]: the 64 MiB --merkle document.  No Hyrax commitment and no consistency argument; the Merkle gadget
    inflates the step circuit (122 lookups x 27 levels per step: a 1 022 173-point Pallas key, 2^20 pre-shifted points), the document is committed by the Poseidon
    tree over 2^27 symbols (row N4, stand-in constants: timing and plumbing; parity of the tree is tests/test_gpu_merkle.py)."""
    from reef_amd import replay
    line = replay.run("cfg5", nofold=True, tables=False)
    shape = next(s for name, s in replay.shapes().items() if "cfg5" in name)
    assert (line["w1"], line["c1"], line["w2"], line["c2"]) == (shape["w1"], shape["c1"], shape["w2"], shape["c2"])
    assert line["key_pallas"] == 1 << 20 and line["w1"] > 1 << 19
    assert line["commitments_checked_against_dlog"] == _expected_checks(shape)
    assert line["consistency_rounds"] == 0 and line["consistency_ipa_ms"] == 0          # --merkle has no Hyrax argument
    assert line["commit_merkle_log"] == 27 and line["commit_merkle_ms"] > 0
    assert line["ipa_pallas_rounds"] == 20 and line["total_prove_msm_ms"] > 0


def test_replay_with_generator_folds(gpu_lib):
    """The zero-patch drop-in path: cross terms over re-keyed contexts + reef_fold per round (the smallest config keeps it short)."""
    from reef_amd import replay
    line = replay.run("cfg1", nofold=False, tables=False)
    shape = next(s for name, s in replay.shapes().items() if "cfg1" in name)
    assert line["commitments_checked_against_dlog"] == _expected_checks(shape)
    assert "generator fold" in line["ipa"]


@pytest.mark.parametrize("cfg,members", [("cfg4", 3), ("cfg3", 2), ("cfg4", 8)])
def test_replay_multi_device_leg_in_one_process(cfg, members, gpu_lib):
    """`reef_replay cfgN nofold devices=M`: the final SNARK's arguments placed whole on M members (per-device contexts, one caller
    thread each; framework.rs:695-721) and the document commitment through a device group (commitment.rs:187), from ONE process --
    what a Rust prover can call.  A test box has one GPU: the ordinals repeat device 0 and the line says so."""
    from reef_amd import replay
    line = replay.run(cfg, nofold=True, devices=[0] * members)
    dv = line["devices"]
    assert dv["members"] == members and dv["distinct_devices"] == 1 and dv["visible_devices"] >= 1
    placed = dv["final_snark_placed"]
    assert [a["argument"] for a in placed] == ["ipa_pallas", "ipa_vesta", "consistency"]
    assert all(a["device"] == 0 and a["alone_ms"] > 0 for a in placed)
    assert placed[0]["points"] == line["key_pallas"] and placed[1]["points"] == line["key_vesta"]
    assert 0 < dv["three_arguments_on_devices_ms"] < dv["three_arguments_one_after_the_other_ms"] * 1.2
    assert dv["commit_rows_checked_against_one_device"] is True and dv["commit_hyrax_group_ms"] > 0
    assert dv["group_exchange"].startswith("peer")
    with pytest.raises(RuntimeError):
        replay.run(cfg, nofold=True, devices=[0, 99])               # an ordinal that is not visible
    with pytest.raises(RuntimeError):
        replay.run("cfg1", nofold=False, devices=[0, 0])            # the leg replays the fold-free final SNARK


def test_replay_multi_device_leg_builds_the_merkle_commitment_in_blocks(gpu_lib):
    """cfg5 (--merkle, 2^27 symbols) with devices=4: the Poseidon tree in four blocks (reef_merkle_commit_devices), its root equal to
    one device's (checked inside the replay); two arguments to place (merkle mode has no consistency IPA, commitment.rs:94-100)."""
    from reef_amd import replay
    line = replay.run("cfg5", nofold=True, devices=[0, 0, 0, 0])
    dv = line["devices"]
    assert [a["argument"] for a in dv["final_snark_placed"]] == ["ipa_pallas", "ipa_vesta"]
    assert dv["commit_merkle_blocks"] == 4 and dv["commit_merkle_root_checked_against_one_device"] is True
    assert dv["commit_merkle_devices_ms"] > 0 and dv["commit_merkle_one_device_ms"] > 0
    assert dv["commit_rows_checked_against_one_device"] is False and dv["commit_hyrax_group_ms"] == 0


def test_replay_executable(gpu_lib):
    """The same harness as a program (what profiles/*replay*.jsonl were recorded with): exit code 0 and one JSON line."""
    exe = os.path.join(ROOT, "reef_amd", "_lib", "reef_replay")
    out = subprocess.run([exe, "cfg3", "nofold"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["commitments_checked_against_dlog"] >= 24
    dev = subprocess.run([exe, "cfg4", "nofold", "devices=4"], capture_output=True, text=True, timeout=600)
    assert dev.returncode == 0, dev.stderr[-2000:]
    assert json.loads(dev.stdout.strip().splitlines()[-1])["devices"]["members"] == 4
    bad = subprocess.run([exe, "cfg3", "nofold", "shapes=/nonexistent.json"], capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "shapes" in bad.stderr
