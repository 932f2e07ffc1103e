"""Row N1 oracle (oracle/keygen_oracle.py): what can be pinned without nova-snark / pasta_curves in the tree.

  * expand_message_xmd against the SHA-256 vectors of RFC 9380 appendix K.1 (the construction is generic over H; the
    product uses H = BLAKE2b-512, hashlib's in the oracle);
  * the maps through their defining properties: SSWU lands on E', the isogeny lands on y^2 = x^3 + 5, is a group
    homomorphism and has a kernel of order 3, the outputs have the (prime) group order;
  * Z satisfies the four conditions of RFC 9380 6.6.2.
Which (E', isogeny, Z, DST) pasta_curves fixes is NOT pinned [R]: those are inputs of the product call."""
import hashlib
import random

import pytest

from oracle import keygen_oracle as K
from oracle import pasta_oracle as O

CURVES = ("pallas", "vesta")


def test_expand_message_xmd_rfc9380_k1_vectors():
    dst = b"QUUX-V01-CS02-with-expander-SHA256-128"
    assert K.expand_message_xmd(b"", dst, 0x20, "sha256", 32, 64).hex() == "68a985b87eb6b46952128911f2a4412bbc302a9d759667f87f7a21d803f07235"
    assert K.expand_message_xmd(b"abc", dst, 0x20, "sha256", 32, 64).hex() == "d8ccab23b5985ccea865c6c97b6e5b8350e794e603b4b97902f53a8a0d605615"
    out = K.expand_message_xmd_blake2b(b"abc", b"dst", 128)
    assert len(out) == 128 and out[:64] != out[64:]


def test_shake_chunks_are_a_prefix_stream():
    a, b = K.shake256_chunks(b"ck", 3), K.shake256_chunks(b"ck", 7)
    assert a == b[:3] and b"".join(b) == hashlib.shake_256(b"ck").digest(224)


@pytest.mark.parametrize("curve", CURVES)
def test_sqrt(curve):
    p = O.CURVES[curve].base
    rng = random.Random(5)
    for _ in range(20):
        x = rng.randrange(p)
        r = K.sqrt_mod(x * x % p, p)
        assert r is not None and r * r % p == x * x % p
    assert K.sqrt_mod(0, p) == 0 and K.sqrt_mod(5, p) is None        # 5 generates the multiplicative group of both fields


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("root", (0, 1, 2))
def test_standin_parameters_are_valid(curve, root):
    k = K.standin_params(curve, root)
    p = k.p
    g = lambda x: (x * x * x + k.a * x + k.b) % p
    assert k.a and k.b
    # Z: non-square, != -1, g(x) - Z irreducible (checked inside find_z), g(B / (Z A)) square
    assert pow(k.z, (p - 1) // 2, p) == p - 1 and k.z != p - 1
    assert K.sqrt_mod(g(k.b * pow(k.z * k.a, -1, p) % p), p) is not None
    rng = random.Random(root)
    pts = []
    for _ in range(12):
        u = rng.randrange(p)
        x, y = K.sswu(u, k)
        assert (y * y - g(x)) % p == 0                                 # on E'
        assert K.sgn0(u) == K.sgn0(y)
        img = K.iso_map((x, y), k)
        assert img is not None and (img[1] ** 2 - img[0] ** 3 - 5) % p == 0   # on y^2 = x^3 + 5
        pts.append(((x, y), img))
    # homomorphism: iso(P + Q) = iso(P) + iso(Q), iso(2P) = 2 iso(P)
    for (P1, I1), (P2, I2) in zip(pts, pts[1:]):
        assert K.iso_map(K.add_general(P1, P2, k.a, p), k) == K.add_general(I1, I2, 0, p)
        assert K.iso_map(K.add_general(P1, P1, k.a, p), k) == K.add_general(I1, I1, 0, p)
    # exceptional inputs of the map: u = 0 and the u with tv2 = 0 need no special treatment by the caller
    x, y = K.sswu(0, k)
    assert (y * y - g(x)) % p == 0


@pytest.mark.parametrize("curve", CURVES)
def test_from_label_points(curve):
    k = K.standin_params(curve)
    cv = O.CURVES[curve]
    pts = K.from_label(b"reef-test-label", 6, k)
    assert len(set(pts)) == 6
    for P in pts:
        assert (P[1] ** 2 - P[0] ** 3 - 5) % k.p == 0
        assert cv.mul(cv.order - 1, P) == cv.neg(P)                    # of the (prime) group order
    assert K.from_label(b"reef-test-label", 3, k) == pts[:3]           # a longer key extends a shorter one
    assert K.from_label(b"other", 1, k)[0] != pts[0]
    le = K.standin_params(curve, little_endian=True)
    assert K.from_label(b"reef-test-label", 1, le)[0] != pts[0]


def test_library_shake256_is_fips202():
    """The product's own SHAKE256 (csrc/keccak.h, host code of libreef_msm.so) against hashlib, across the rate boundary."""
    from reef_amd import keygen
    for msg, n in ((b"", 32), (b"ck", 4096), (b"x" * 135, 137), (b"y" * 136, 272), (b"z" * 1000, 1), (b"w" * 137, 136)):
        assert keygen.shake256(msg, n) == hashlib.shake_256(msg).digest(n)


def test_oracle_matches_committed_fixture():
    """tests/golden/next_rows_golden.json (oracle/gen_golden_next_rows.py): the oracle's outputs on the stand-in parameter
    sets, committed so that a change of the oracle shows up here (regression anchors, not reference vectors)."""
    import json
    import os
    data = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "next_rows_golden.json")))
    for case in data["keygen"]:
        k = K.standin_params(case["curve"], case["root_index"], case["little_endian"])
        assert hex(k.a) == case["a"] and hex(k.z) == case["z"]
        pts = K.from_label(bytes.fromhex(case["label_hex"]), len(case["points"]), k)
        assert [[hex(p[0]), hex(p[1])] for p in pts] == case["points"]
