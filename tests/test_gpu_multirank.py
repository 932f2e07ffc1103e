"""The N > 1 path of bench.py with TWO REAL RANKS: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2`,
exactly as the driver launches it, on the one GPU a test box has (the ranks share device 0; the 96-byte partial sums
are exchanged through gloo, which bench.py labels as the host-staged fallback -- RCCL needs one device per rank).
Both splits: by points (every rank owns its own 2^logn pairs, weak scaling) and by Pippenger window (one MSM, rank r
accumulates the windows w = r mod 2 through reef_msm_ctx_set_window_split, strong scaling).  Every rank checks the
COMBINED point against the discrete-log closed form of the whole MSM, and that its own partial is not already the total.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(sharding, logn, extra=()):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--sharding", sharding,
           "--logn", str(logn), "--steps", "6", "--warmup", "2", "--no-cpu-baseline", *extra]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("sharding,logn", [("points", 16), ("windows", 16), ("windows", 18)])
def test_bench_with_two_ranks(sharding, logn, gpu_lib):
    line = _run(sharding, logn)
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["check"] == "dlog-ok"
    assert cfg["partials_differ_from_total"] is True             # each rank really held a partial sum
    assert "gloo" in cfg["exchange"]                             # labelled as the host-staged fallback, not as RCCL
    assert cfg["rccl"] == {"ranks_seen": 2, "backend": "gloo", "stream_ordered": False}
    assert cfg["streams_of_the_contexts"] == cfg["streams"]       # contexts whose stream is handed out sit on streams of their own
    if sharding == "points":
        assert line["scaling"] == "weak" and cfg["total_points"] == 2 << logn and cfg["sharding"] == "points"
        # one SCALE run yields both strong splits as well: ONE 2^logn-point MSM by window and by points on the same ranks,
        # both combined points checked against the discrete log of the whole MSM (inside cfg["check"])
        ss = cfg["strong_scaling"]
        assert ss["one_msm_points"] == 1 << logn and ss["windows_ms_per_step"] > 0 and ss["points_ms_per_step"] > 0
        ph = ss["phases"]                                      # round 6: the split itemised per rank beside the model's prediction, so that the first SCALE line explains itself
        for split in ("windows", "points"):
            assert len(ph[split]["member_msm_ms"]["per_rank"]) == 2 and ph[split]["member_msm_ms"]["max"] >= ph[split]["member_msm_ms"]["min"] > 0
            assert ph[split]["exchange_ms"]["max"] > 0
        assert ph["predicted"]["windows_member_msm_ms"] > 0 and "DESIGN.md 7" in ph["predicted"]["basis"]
        hy = ss["hyrax_rows"]                                  # the row-sharded Hyrax commitment of BASELINE configs[3]
        assert hy["rows"] == 4096 and hy["rows_per_rank"] == 2048 and hy["check"] == "dlog-ok" and hy["ms_per_commit"] > 0
        fs, scs = ss["final_snark"], ss["sumcheck"]            # round 4: whole units over the ranks
        assert fs["owner"] == [0, 1, 1] and fs["check"] == "same-points" and fs["ms"] > 0 and fs["one_gpu_one_after_the_other_ms"] > 0
        assert scs["ell"] == 26 and scs["entries_per_rank"] == 1 << 25 and scs["check"] == "sumcheck-identity-ok" and scs["ms_per_step"] > 0
        sp = ss["single_process"]                              # round 5: the same devices driven by ONE process through reef_msm_group_* (a child of rank 0)
        assert sp["check"] == "dlog-ok" and sp["devices"] == [0, 0] and sp["exchange"].startswith("peer") and sp["value"] > 0
        assert sp["strong_scaling"]["windows_ms_per_step"] > 0 and sp["strong_scaling"]["points_ms_per_step"] > 0
        spr = ss["single_process_rccl"]                        # and with the partial sums as ncclSend/ncclRecv on a single-process communicator
        assert spr["check"] == "dlog-ok" and spr["exchange"].startswith("rccl") and spr["value"] > 0 and spr["strong_scaling"]["windows_ms_per_step"] > 0
        assert set(ss["speedup_vs_1"]) == {"windows", "points"} and ss["one_gpu_ms_per_msm"] == pytest.approx(cfg["ms_per_msm"]) and cfg["msms_per_step"] == 256 \
            and line["ms_per_step"] == pytest.approx(256 * cfg["ms_per_msm"])
    else:
        assert line["scaling"] == "strong" and cfg["total_points"] == 1 << logn and cfg["sharding"].startswith("windows")
        assert cfg["strong_scaling"] is None
    assert line["value"] > 0 and line["roofline"]["kernel_ms"] > 0


def test_side_legs_under_a_deadline(gpu_lib):
    """The legs after the timed region (strong splits, whole units, the single-process child) have never met two physical GPUs: if one of
    their collectives hangs, the line of the completed timed region must still be printed.  With a one-second deadline the watchdog fires in
    the middle of the strong legs: rank 0 prints the line it has -- the weak region's value, already checked by discrete logarithm -- and
    every rank leaves with exit code 0."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--logn", "16", "--steps", "4",
           "--warmup", "1", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", REEF_BENCH_SIDE_TIMEOUT="1"), cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    cfg = line["config"]
    assert cfg["check"].startswith("dlog-ok (timed region") and "REEF_BENCH_SIDE_TIMEOUT" in cfg["strong_scaling"]["error"]
    assert cfg["partials_differ_from_total"] is True and line["value"] > 0 and line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert "did not finish within 1 s" in out.stderr


def test_world_size_must_match_gpus(gpu_lib):
    """--gpus N under a launcher with another world size is refused instead of quietly measuring something else."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--backend", "gloo", "--logn", "12", "--steps", "1",
           "--warmup", "0", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr


@pytest.mark.parametrize("members,logn,exchange", [(2, 16, "peer"), (8, 14, "peer"), (3, 17, "peer"), (3, 14, "rccl"), (2, 14, "host")])
def test_bench_single_process_over_a_device_group(members, logn, exchange, gpu_lib):
    """`bench.py --gpus N --single-process`: no launcher, no torch -- ONE process drives N members through reef_msm_group_* (what a Rust
    prover can call; Reef is one process, src/backend/main.rs:82).  On a one-GPU box the ordinals repeat device 0 and the line says
    that it is not a scaling measurement; every combined point is checked against its discrete logarithm."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(members), "--single-process", "--logn", str(logn), "--steps", "3", "--warmup", "1",
           "--msms-per-step", "6", "--group-exchange", exchange]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    cfg = line["config"]
    assert line["n_gpus"] == members and cfg["check"] == "dlog-ok" and line["scaling"] == "weak"
    assert cfg["mode"].startswith("single-process") and cfg["devices"] == [0] * members and cfg["distinct_devices"] == 1
    assert "NOT a scaling measurement" in cfg["devices_note"] if members > 1 else cfg["devices_note"] is None
    assert cfg["exchange"].startswith({"peer": "peer", "rccl": "rccl", "host": "host-staged"}[exchange]) and cfg["total_points"] == members << logn
    assert sum(cfg["key_points_per_member"]) == members << logn
    ss = cfg["strong_scaling"]
    assert ss["one_msm_points"] == 1 << logn and set(ss["speedup_vs_1"]) == {"windows", "points"}
    for k in ("windows_ms_per_step", "points_ms_per_step", "windows_latency_ms", "points_latency_ms", "one_gpu_ms_per_msm"):
        assert ss[k] > 0, k
    ph = ss["phases"]                                          # round 6: where one split call's time goes (reef_msm_group_last_timing), both scalar routes
    for key in ("windows", "windows_fanout_from_member_0", "points"):
        h = ph[key]["host_scalars"]
        assert "error" not in ph[key] and h["total_ms"] > 0 and h["scalar_distribution_ms"] > 0 and h["exchange_ms"] > 0
        assert h["member_msm_ms"]["max"] >= h["member_msm_ms"]["min"] > 0
    assert ph["windows"]["device_scalars"]["scalar_distribution_ms"] < ph["windows"]["host_scalars"]["scalar_distribution_ms"]
    assert ph["predicted"]["windows_latency_ms"] > 0
    if logn >= 16:
        assert ss["hyrax_rows"]["check"] == "dlog-ok" and ss["hyrax_rows"]["rows"] == 4096
        assert ss["merkle_commit"]["check"] == "same-root" and ss["merkle_commit"]["blocks"] == {2: 2, 3: 2}[members] and ss["merkle_commit"]["ms_per_commit"] > 0
    assert line["value"] > 0 and line["roofline"]["kernel_ms"] > 0 and cfg["host_scalars_ms_per_msm"] > 0
    refused = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, WORLD_SIZE="2"))
    assert refused.returncode != 0 and "WITHOUT torch.distributed.run" in refused.stderr
