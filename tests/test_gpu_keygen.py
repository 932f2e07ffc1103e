"""Row N1 on the GPU: reef_derive_generators against oracle/keygen_oracle.py, bit-exact, with the stand-in parameter
sets (every cube root of the isogeny's kernel, both byte orders); key-sized derivation through size-independent
properties (on the curve, distinct, a longer key extends a shorter one) and through an MSM over the derived key."""
import hashlib

import numpy as np
import pytest

from oracle import keygen_oracle as K
from oracle import pasta_oracle as O

pytestmark = pytest.mark.gpu


def _derive(curve, label, n, k):
    from reef_amd import keygen
    raw = keygen.derive_generators(curve, label, n, k.a, k.b, k.z, k.iso, k.dst, k.little_endian)
    return raw, keygen.points_to_ints(curve, raw)


@pytest.mark.parametrize("curve", ("pallas", "vesta"))
@pytest.mark.parametrize("root,le", ((0, False), (1, False), (2, True)))
def test_generators_match_oracle(curve, root, le, gpu_lib):
    k = K.standin_params(curve, root, le)
    for label, n in ((b"ck", 1), (b"reef test label", 67), (b"", 5)):
        _, got = _derive(curve, label, n, k)
        assert got == K.from_label(label, n, k)


def test_library_shake256_matches_hashlib(gpu_lib):
    from reef_amd import keygen
    for msg, n in ((b"", 32), (b"ck", 4096), (b"x" * 135, 137), (b"y" * 136, 272), (b"z" * 1000, 1)):
        assert keygen.shake256(msg, n) == hashlib.shake_256(msg).digest(n)


def test_key_sized_derivation_properties(gpu_lib):
    curve, n = "pallas", 1 << 17                                       # the size of Reef's largest keys
    k = K.standin_params(curve)
    raw, pts = _derive(curve, b"ck", n, k)
    p = k.p
    xs = np.array([pt[0] % (1 << 64) for pt in pts], dtype=np.uint64)
    assert all(pt is not None for pt in pts) and len(np.unique(xs)) == n
    for i in (0, 1, 77, 4095, 65536, n - 1):
        assert (pts[i][1] ** 2 - pts[i][0] ** 3 - 5) % p == 0
    for i in (0, 1, 77, n - 1):                                        # sampled against the oracle
        assert pts[i] == K.hash_to_curve(K.shake256_chunks(b"ck", n)[i], k)
    _, short = _derive(curve, b"ck", 100, k)
    assert short == pts[:100]
    # the derived key is a valid MSM key: sum of the first 64 generators through the engine = oracle sum
    from reef_amd import msm
    cv = O.CURVES[curve]
    ctx = msm.MsmContext(curve, raw[:64])
    ones = np.zeros((64, 4), dtype=np.uint64)
    ones[:, 0] = 1
    got = np.asarray(ctx.msm(ones, is_mont=False)).reshape(12)
    acc = None
    for pt in pts[:64]:
        acc = cv.add(acc, pt)
    rinv = pow(1 << 256, -1, p)
    X, Y, Z = (sum(int(got[4 * c + j]) << (64 * j) for j in range(4)) * rinv % p for c in range(3))
    zi = pow(Z, -1, p)
    assert (X * zi * zi % p, Y * zi * zi * zi % p) == acc


def test_gpu_matches_committed_fixture(gpu_lib):
    import json
    import os
    from reef_amd import keygen
    data = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "next_rows_golden.json")))
    for case in data["keygen"]:
        k = K.standin_params(case["curve"], case["root_index"], case["little_endian"])
        raw = keygen.derive_generators(case["curve"], bytes.fromhex(case["label_hex"]), len(case["points"]), k.a, k.b, k.z, k.iso, k.dst, k.little_endian)
        got = keygen.points_to_ints(case["curve"], raw)
        assert [[hex(p[0]), hex(p[1])] for p in got] == case["points"]


@pytest.mark.parametrize("curve", ("pallas", "vesta"))
def test_commitment_gens_new_from_a_label(curve, gpu_lib):
    """CommitmentGens::new(label, n) / new_with_blinding_gen as Reef calls them (framework.rs:297-303, commitment.rs:176-180) through the
    provider mirror: the generators are the oracle's from_label, and a commitment over them (with and without a blind) is the oracle's."""
    from oracle import pasta_ref as R
    from reef_amd import keygen, provider
    k = K.standin_params(curve)
    cid = 0 if curve == "pallas" else 1
    n = 300
    kw = dict(a=k.a, b=k.b, z=k.z, iso=k.iso, dst=k.dst, little_endian=k.little_endian)
    gens = provider.CommitmentGens.new(curve, b"reef ck", n, **kw)
    assert len(gens) == n and keygen.points_to_ints(curve, gens.bases) == K.from_label(b"reef ck", n, k)
    v = R.gen_scalars(cid, 99, n, kind=1)
    assert gens.commit(v).compress() == R.compress(cid, R.msm_pippenger(cid, gens.bases, v))
    h = R.gen_bases_ap(cid, 0xB11D, 1, 1)[0].copy()
    blind = R.gen_scalars(cid, 100, 1)
    gb = provider.CommitmentGens.new_with_blinding_gen(curve, b"reef ck", n, h, **kw)
    assert np.array_equal(gb.bases, gens.bases)
    assert gb.commit(v, blind).compress() == R.compress(cid, R.row_msm(cid, gens.bases, v, 1, n, h=h, blinds=blind))
    gens.close()
    gb.close()
