"""The oracle against the golden vectors and against itself (big-int Python vs C restatement).

The reference holds no MSM known-answer vectors: the anchors pinned here are the
published curve definition (generator (-1,2), group orders) and the values listed in SURVEY.md.
"""
import numpy as np
import pytest

from oracle.pasta_oracle import CURVES, P, Q, SplitMix64, ap_bases, msm_via_dlog, uniform_scalar

CID = {"pallas": 0, "vesta": 1}

SURVEY_ANCHORS = {
    ("pallas", 2): "030000b067c50313fcac1144eee2fe0e0000000000000000000000000000001c",
    ("pallas", 3): "63d232eb3b8af0b75cfcf55ade47f6ff4cdf4e47a7454cb8ed67a9ba6f56e788",
    ("pallas", 5): "d10e70fdf461fb465db10c602adbd7b3fd9fdb0d492d1ecd4cbdffedecaa0ab3",
    ("pallas", Q - 1): "00000000ed302d991bf94c09fc984622000000000000000000000000000000c0",
    ("pallas", 100): "f52ccaaf588ca0cc5f220c7085b97e22bc92a3faa76ab78ef92e1d7a108c5b3d",
}


def test_moduli_match_reference_constant():
    # src/backend/r1cs_helper.rs:37-38 hard-codes the Pallas scalar modulus in decimal
    assert Q == 28948022309329048855892746252171976963363056481941647379679742748393362948097
    assert P == 28948022309329048855892746252171976963363056481941560715954676764349967630337


@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_group_structure(name):
    C = CURVES[name]
    assert C.is_on_curve(C.gen)
    assert C.mul(C.order, C.gen) is None
    assert C.add(C.mul(C.order - 1, C.gen), C.gen) is None
    assert pow(C.base, 1, 3) == 1  # GLV endomorphism exists (not used yet)


def test_survey_anchors():
    C = CURVES["pallas"]
    for (name, k), hexv in SURVEY_ANCHORS.items():
        assert C.compress(C.mul(k, C.gen)).hex() == hexv
    V = CURVES["vesta"]
    assert V.mul(2, V.gen)[0] == 0x1C0000000000000000000000000000000EFEE2EE443109E0ED5F06DE70000003


def test_golden_anchors(golden):
    for a in golden["anchors"]:
        C = CURVES[a["curve"]]
        pt = C.mul(int(a["k"], 16), C.gen)
        assert C.compress(pt).hex() == a["compressed"]
        assert C.decompress(bytes.fromhex(a["compressed"])) == pt


def test_golden_explicit_python(golden):
    for case in golden["explicit"]:
        C = CURVES[case["curve"]]
        sc = [C.scalar_from_mont(int.from_bytes(bytes.fromhex(h), "little")) for h in case["scalars_mont_hex"]]
        sc2 = [int.from_bytes(bytes.fromhex(h), "little") for h in case["scalars_canon_hex"]]
        assert sc == sc2
        bases = [C.affine_from_bytes(bytes.fromhex(h)) for h in case["bases_hex"]]
        assert C.compress(C.msm_naive(sc, bases)).hex() == case["expect_compressed"], case["label"]


def _explicit_arrays(case):
    n = len(case["bases_hex"])
    bases = np.frombuffer(b"".join(bytes.fromhex(h) for h in case["bases_hex"]), dtype=np.uint64).reshape(n, 8).copy() if n else np.zeros((0, 8), np.uint64)
    sm = np.frombuffer(b"".join(bytes.fromhex(h) for h in case["scalars_mont_hex"]), dtype=np.uint64).reshape(n, 4).copy() if n else np.zeros((0, 4), np.uint64)
    sc = np.frombuffer(b"".join(bytes.fromhex(h) for h in case["scalars_canon_hex"]), dtype=np.uint64).reshape(n, 4).copy() if n else np.zeros((0, 4), np.uint64)
    return bases, sm, sc


def test_golden_explicit_c(golden, cref):
    for case in golden["explicit"]:
        cid = CID[case["curve"]]
        bases, sm, sc = _explicit_arrays(case)
        for scal, mont in ((sm, True), (sc, False)):
            for fn in (cref.msm_naive, lambda *a, **k: cref.msm_pippenger(*a, threads=2, **k),
                       lambda *a, **k: cref.msm_pippenger_windows(*a, threads=3, **k)):
                r = fn(cid, bases, scal, mont=mont)
                assert cref.compress(cid, r).hex() == case["expect_compressed"], case["label"]


def test_golden_seeded_c(golden, cref):
    import hashlib
    for case in golden["seeded"]:
        cid = CID[case["curve"]]
        n = case["n"]
        bases = cref.gen_bases_ap(cid, case["k0"], case["d"], n)
        sc = cref.gen_scalars(cid, case["seed"], n, kind=case["kind"])
        assert hashlib.sha256(bases.tobytes() + sc.tobytes()).hexdigest() == case["input_sha256"]
        r = cref.msm_pippenger(cid, bases, sc, threads=4)
        assert cref.compress(cid, r).hex() == case["expect_compressed"], (case["curve"], n, case["kind"])
        for threads in (1, 5):   # the window-parallel form on the thread pool (the timed cpu_baseline)
            r = cref.msm_pippenger_windows(cid, bases, sc, threads=threads)
            assert cref.compress(cid, r).hex() == case["expect_compressed"], (case["curve"], n, case["kind"], threads)


@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_c_field_ops_vs_bigint(name, cref):
    C = CURVES[name]
    f = CID[name]  # coordinate field of curve
    m = C.base
    rng = SplitMix64(77)
    R = 1 << 256
    for _ in range(50):
        a, b = uniform_scalar(rng, m), uniform_scalar(rng, m)
        A, B = cref.int_to_limbs(a), cref.int_to_limbs(b)
        assert cref.limbs_to_int(cref.field_op("fmul", f, A, B)) == a * b * pow(R, -1, m) % m
        assert cref.limbs_to_int(cref.field_op("fadd", f, A, B)) == (a + b) % m
        assert cref.limbs_to_int(cref.field_op("fsub", f, A, B)) == (a - b) % m
        assert cref.limbs_to_int(cref.field_op("to_mont", f, A)) == a * R % m
        assert cref.limbs_to_int(cref.field_op("from_mont", f, A)) == a * pow(R, -1, m) % m
        if a:
            # finv works on Montgomery form: inv(aR) = a^-1 R
            am = cref.int_to_limbs(a * R % m)
            assert cref.limbs_to_int(cref.field_op("finv", f, am)) == pow(a, -1, m) * R % m


def test_c_fold_and_rows_vs_golden(golden, cref):
    for case in golden["fold"]:
        cid = CID[case["curve"]]
        gens = cref.gen_bases_ap(cid, case["k0"], case["d"], case["n"])
        out = cref.fold(cid, gens, int(case["w1"], 16), int(case["w2"], 16))
        jac = np.zeros((out.shape[0], 12), dtype=np.uint64)
        one = cref.field_op("to_mont", cid, cref.int_to_limbs(1))
        jac[:, :8] = out
        jac[:, 8:] = one
        comp = cref.compress(cid, jac)
        assert [comp[32 * i:32 * i + 32].hex() for i in range(out.shape[0])] == case["expect_compressed"]
    for case in golden["rows"]:
        cid = CID[case["curve"]]
        C = CURVES[case["curve"]]
        rows, row_len = case["rows"], case["row_len"]
        bases = cref.gen_bases_ap(cid, case["k0"], case["d"], row_len)
        sc = np.array([[s, 0, 0, 0] for s in case["scalars"]], dtype=np.uint64)
        bl = np.array([cref.int_to_limbs(int(b, 16)) for b in case["blinds"]], dtype=np.uint64)
        h = np.frombuffer(C.affine_to_bytes(C.mul(case["h_k"], C.gen)), dtype=np.uint64).copy()
        r = cref.row_msm(cid, bases, sc, rows, row_len, h=h, blinds=bl, mont=False, threads=2)
        comp = cref.compress(cid, r)
        assert [comp[32 * i:32 * i + 32].hex() for i in range(rows)] == case["expect_compressed"]


def test_dlog_property_c_large(cref):
    """Size-independent check used at BASELINE sizes: MSM over (k0+i*d)G == (sum s_i (k0+i d)) G."""
    n = 1 << 14
    for cid, name in ((0, "pallas"), (1, "vesta")):
        C = CURVES[name]
        bases = cref.gen_bases_ap(cid, 11, 7, n)
        sc = cref.gen_scalars(cid, 4242, n, kind=1)
        r = cref.msm_pippenger(cid, bases, sc, threads=4)
        canon = [C.scalar_from_mont(cref.limbs_to_int(sc[i])) for i in range(n)]
        assert cref.compress(cid, r) == C.compress(msm_via_dlog(C, canon, 11, 7))
        assert cref.compress(cid, cref.msm_pippenger_windows(cid, bases, sc, threads=8)) == cref.compress(cid, r)
        assert cref.compress(cid, cref.msm_pippenger_windows(cid, bases, sc, threads=2, n=n - 37)) == \
            cref.compress(cid, cref.msm_pippenger(cid, bases[:n - 37].copy(), sc[:n - 37].copy(), threads=2))


def test_c_threaded_fold_equals_serial(cref):
    for cid in (0, 1):
        gens = cref.gen_bases_ap(cid, 5, 9, 70)
        w1, w2 = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF, 0x0FEDCBA987654321
        assert (cref.fold_mt(cid, gens, w1, w2, threads=4) == cref.fold(cid, gens, w1, w2)).all()


def test_window_plan_keeps_the_window_of_the_whole_input():
    from oracle import pasta_ref
    c1, s1 = pasta_ref.window_plan(1 << 20, 1)
    c256, s256 = pasta_ref.window_plan(1 << 20, 256)
    assert c1 >= 14 and c256 >= 12 and s256 * ((256 + c256 - 1) // c256) >= 128   # many threads: tiles, not a smaller window
