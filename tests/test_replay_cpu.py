"""The CPU leg bench.py reports beside the GPU replay (oracle/replay_cpu.py): Reef's MSM sequence through the oracle's C
restatement -- here on the smallest config, to see that it runs the sequence it says and that its pieces are the oracle's."""
import os

from oracle import pasta_ref as R
from oracle import replay_cpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = os.path.join(ROOT, "tests", "golden", "replay_shapes.json")


def test_cpu_replay_of_the_smallest_config():
    sh = replay_cpu.load_shape("cfg1", SHAPES)
    out = replay_cpu.run("cfg1", SHAPES, threads=4)
    assert out["threads"] == 4 and "NOT the reference binary" in out["kind"]
    assert out["fold_steps_ms"] > 0 and out["ms_per_step"] == out["fold_steps_ms"] / sh["steps"]
    assert out["ipa_pallas_ms"] > 0 and out["ipa_vesta_ms"] > 0 and out["consistency_ipa_ms"] > 0
    assert abs(out["total_prove_msm_ms"] - (out["fold_steps_ms"] + out["final_snark_ms"] + out["consistency_ipa_ms"])) < 1e-6


def test_ipa_round_pieces_are_the_oracles():
    """One round as replay_cpu issues it: two cross-term MSMs over halves of the generators and the joint fold."""
    n = 64
    gens = R.gen_bases_ap(0, 5, 3, n)
    sc = R.gen_scalars(0, 7, n)
    half = n // 2
    lo, hi = gens[:half], gens[half:]
    assert R.compress(0, R.msm_pippenger_windows(0, hi, sc[:half], threads=2)) == R.compress(0, R.msm_naive(0, hi.copy(), sc[:half].copy()))
    w1, w2 = 0x1234567, 0x89ABCDEF012345
    assert (R.fold_mt(0, gens, w1, w2, 2, half=half) == R.fold(0, gens, w1, w2)).all()
