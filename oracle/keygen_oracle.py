"""CPU restatement of commitment-key derivation (row N1) -- TEST INFRASTRUCTURE, never imported by the product.

What Reef calls (src/backend/framework.rs:297-303, src/backend/commitment.rs:146-149,176-180):
    CommitmentGens::new(label, n) -> from_label(label, n)        [R: nova-snark provider/pedersen.rs, not in the reference tree]
        SHAKE256(label) squeezed into n 32-byte strings, each mapped with pasta_curves' hash_to_curve("from_uniform_bytes")
        [R: fil_pasta_curves 0.5.2 hashtocurve.rs], the results batch-normalised to affine points.
hash_to_curve is the hash_to_curve of RFC 9380 for the suite <curve>_XMD:BLAKE2b_SSWU_RO_: two field elements by
expand_message_xmd over BLAKE2b-512 (hash_to_field, RFC 9380 5.2-5.3), the simplified SWU map on a curve E' with a*b != 0
that is 3-isogenous to the j = 0 curve (6.6.2-6.6.3), the isogeny back, and the sum of the two images.

PINNED here: SHAKE256 and BLAKE2b are hashlib's; expand_message_xmd against the SHA-256 vectors of RFC 9380 appendix K.1
(tests/test_keygen_oracle.py); hash_to_field, the straight-line SSWU of RFC 9380 appendix F.2 and the isogeny evaluation restated from the RFC and checked through their
defining properties (points on E', images on y^2 = x^3 + 5, the isogeny is a group homomorphism with a kernel of order 3).
UNPINNED [R]: which of the 3 x 6 = 18 valid (E', isogeny) pairs pasta_curves fixes, its Z, its byte order in hash_to_field and
its domain-separation string -- all PARAMETERS of the product call (include/reef_msm.h, reef_keygen_params), taken by
the Rust binding from the crate's constants.  `standin_params` derives ONE valid choice with Velu's formulas:
    E': y^2 = x^3 + a x + 1265,  a = -(10/3) x0^2,  x0^3 = 540  (kernel point (x0, sqrt 5)),
    (X, Y) = (x + t/(x-x0) + u/(x-x0)^2, y (1 - t/(x-x0)^2 - 2u/(x-x0)^3)),  t = -(2/3) x0^2, u = 20,  lands on Y^2 = X^3 + 3645,
    and (X/9, Y/27) lies on y^2 = x^3 + 5.
"""
from __future__ import annotations

import hashlib
from typing import List, Optional, Sequence, Tuple

from .pasta_oracle import CURVES

Point = Optional[Tuple[int, int]]


def shake256_chunks(label: bytes, n: int) -> List[bytes]:
    raw = hashlib.shake_256(label).digest(32 * n)
    return [raw[32 * i:32 * i + 32] for i in range(n)]


def expand_message_xmd(msg: bytes, dst: bytes, len_in_bytes: int, hash_name: str = "blake2b", b_in: int = 64, s_in: int = 128) -> bytes:
    """RFC 9380 5.3.1; defaults: H = BLAKE2b-512 (b_in_bytes = 64, s_in_bytes = 128).  Generic over H so that the RFC's own
    SHA-256 vectors (appendix K.1) pin the construction."""
    ell = -(-len_in_bytes // b_in)
    assert ell <= 255 and len_in_bytes <= 65535 and len(dst) <= 255
    dst_prime = dst + bytes([len(dst)])
    h = lambda data: hashlib.new(hash_name, data).digest()
    b0 = h(bytes(s_in) + msg + len_in_bytes.to_bytes(2, "big") + b"\x00" + dst_prime)
    b = [h(b0 + b"\x01" + dst_prime)]
    for i in range(2, ell + 1):
        b.append(h(bytes(x ^ y for x, y in zip(b0, b[-1])) + bytes([i]) + dst_prime))
    return b"".join(b)[:len_in_bytes]


def expand_message_xmd_blake2b(msg: bytes, dst: bytes, len_in_bytes: int) -> bytes:
    return expand_message_xmd(msg, dst, len_in_bytes)


def hash_to_field(msg: bytes, dst: bytes, p: int, count: int = 2, L: int = 64, little_endian: bool = False) -> List[int]:
    """RFC 9380 5.2 (m = 1): count elements from L uniform bytes each (big-endian per the RFC; the flag is the [R] knob)."""
    uniform = expand_message_xmd_blake2b(msg, dst, count * L)
    return [int.from_bytes(uniform[L * i:L * i + L], "little" if little_endian else "big") % p for i in range(count)]


def sqrt_mod(x: int, p: int) -> Optional[int]:
    """Tonelli-Shanks; None for a non-residue."""
    x %= p
    if x == 0:
        return 0
    if pow(x, (p - 1) // 2, p) != 1:
        return None
    s, q = 0, p - 1
    while q % 2 == 0:
        s, q = s + 1, q // 2
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, r = s, pow(z, q, p), pow(x, q, p), pow(x, (q + 1) // 2, p)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2, i = t2 * t2 % p, i + 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c = i, b * b % p
        t, r = t * c % p, r * b % p
    return r


class KeygenParams:
    """Everything of pasta_curves' hash_to_curve that this repository cannot confirm, as data."""

    def __init__(self, p: int, a: int, b: int, z: int, iso: Sequence[int], dst: bytes, little_endian: bool = False):
        self.p, self.a, self.b, self.z, self.iso, self.dst, self.little_endian = p, a % p, b % p, z % p, [c % p for c in iso], dst, little_endian
        # iso: 13 coefficients, highest degree first:  x_num[4], x_den[2] (monic x^2 + ..), y_num[4], y_den[3] (monic x^3 + ..)


def sgn0(x: int) -> int:
    return x & 1


def sswu(u: int, k: KeygenParams) -> Tuple[int, int]:
    """Simplified SWU, straight-line form of RFC 9380 appendix F.2 -> affine point on E': y^2 = x^3 + a x + b."""
    p, A, B, Z = k.p, k.a, k.b, k.z
    tv1 = Z * u * u % p
    tv2 = (tv1 * tv1 + tv1) % p
    tv3 = B * (tv2 + 1) % p
    tv4 = A * (Z if tv2 == 0 else (-tv2) % p) % p
    tv6 = tv4 * tv4 % p
    gx_num = (tv3 * tv3 + A * tv6) % p * tv3 % p
    tv6 = tv6 * tv4 % p
    gx_num = (gx_num + B * tv6) % p                       # g(x1) = gx_num / tv6
    x = tv1 * tv3 % p
    ratio = gx_num * pow(tv6, -1, p) % p
    y1 = sqrt_mod(ratio, p)
    if y1 is not None:
        xx, y = tv3, y1
    else:
        y1 = sqrt_mod(Z * ratio % p, p)                   # then Z * g(x1) is a square: g(x2) = Z^3 u^6 g(x1), y2 = Z u^3 y1 ... as in F.2
        assert y1 is not None
        xx, y = x, tv1 * u % p * y1 % p
    if sgn0(u) != sgn0(y):
        y = (-y) % p
    xx = xx * pow(tv4, -1, p) % p
    return xx, y


def iso_map(pt: Tuple[int, int], k: KeygenParams) -> Point:
    x, y = pt
    p, c = k.p, k.iso
    xn = ((c[0] * x + c[1]) * x + c[2]) * x + c[3]
    xd = (x + c[4]) * x + c[5]
    yn = ((c[6] * x + c[7]) * x + c[8]) * x + c[9]
    yd = ((x + c[10]) * x + c[11]) * x + c[12]
    if xd % p == 0 or yd % p == 0:
        return None                                       # a kernel point
    return xn * pow(xd, -1, p) % p, y * yn % p * pow(yd, -1, p) % p


def add_general(P1: Point, P2: Point, a: int, p: int) -> Point:
    if P1 is None: return P2
    if P2 is None: return P1
    (x1, y1), (x2, y2) = P1, P2
    if x1 == x2 and (y1 + y2) % p == 0: return None
    lam = (3 * x1 * x1 + a) * pow(2 * y1, -1, p) % p if P1 == P2 else (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    return x3, (lam * (x1 - x3) - y1) % p


def hash_to_curve(msg: bytes, k: KeygenParams) -> Point:
    u0, u1 = hash_to_field(msg, k.dst, k.p, 2, 64, k.little_endian)
    q0, q1 = iso_map(sswu(u0, k), k), iso_map(sswu(u1, k), k)
    return add_general(q0, q1, 0, k.p)


def from_label(label: bytes, n: int, k: KeygenParams) -> List[Point]:
    return [hash_to_curve(c, k) for c in shake256_chunks(label, n)]


def find_z(p: int, a: int, b: int) -> int:
    """find_z_sswu of RFC 9380 appendix H.2."""
    def g(x): return (x * x * x + a * x + b) % p
    def is_square(x): return x % p == 0 or pow(x, (p - 1) // 2, p) == 1
    def irreducible_cubic_minus(z):                      # g(x) - z has no root <=> irreducible (degree 3)
        # x^p mod (g - z): gcd with x^p - x is 1 iff no roots; computed by modular exponentiation of polynomials of degree < 3
        def mulmod(f, h):
            r = [0] * 5
            for i, fi in enumerate(f):
                for j, hj in enumerate(h):
                    r[i + j] = (r[i + j] + fi * hj) % p
            for d in (4, 3):                              # x^3 = -a x - (b - z)
                c = r[d]
                r[d] = 0
                r[d - 2] = (r[d - 2] - c * a) % p
                r[d - 3] = (r[d - 3] - c * (b - z)) % p
            return r[:3]
        res, base, e = [1, 0, 0], [0, 1, 0], p
        while e:
            if e & 1: res = mulmod(res, base)
            base, e = mulmod(base, base), e >> 1
        d = [(res[0]) % p, (res[1] - 1) % p, res[2] % p]   # x^p - x mod (g - z)
        # gcd(g - z, d) is non-trivial iff they share a root; a cubic with no root in F_p is irreducible.  Resultant test via roots count:
        # simple way: d == 0 means all three roots in F_p; otherwise test gcd degree by Euclid
        f = [(b - z) % p, a % p, 0, 1]
        def deg(q):
            while q and q[-1] % p == 0: q = q[:-1]
            return len(q) - 1, q
        da, fa = deg(list(f)); db, fb = deg(list(d))
        while db >= 0:
            while da >= db and db >= 0:
                coef = fa[-1] * pow(fb[-1], -1, p) % p
                sh = da - db
                for i in range(db + 1): fa[i + sh] = (fa[i + sh] - coef * fb[i]) % p
                da, fa = deg(fa)
            da, fa, db, fb = db, fb, da, fa
        return da == 0                                    # gcd is a constant
    ctr = 1
    while True:
        for z in (ctr, p - ctr):
            if is_square(z) or z == p - 1: continue
            if not irreducible_cubic_minus(z): continue
            if is_square(g(b * pow(z * a, -1, p) % p)): return z
        ctr += 1


def cube_roots(v: int, p: int) -> List[int]:
    """All cube roots of v mod p (p = 1 mod 3), sorted; [] when v is not a cube."""
    s, m = 0, p - 1
    while m % 3 == 0:
        s, m = s + 1, m // 3
    g = 2
    while pow(pow(g, m, p), 3 ** (s - 1), p) == 1:
        g += 1
    z3, order = pow(g, m, p), 3 ** s                      # a generator of the 3-Sylow subgroup
    y = pow(v, pow(3, -1, m), p)                          # y^3 = v * (an element of the 3-Sylow subgroup)
    corr = v * pow(y, -3, p) % p
    lg = next(i for i in range(order) if pow(z3, i, p) == corr)
    if lg % 3:
        return []
    y = y * pow(z3, lg // 3, p) % p
    om = pow(z3, order // 3, p)
    return sorted([y, y * om % p, y * om * om % p])


def standin_params(curve: str = "pallas", root_index: int = 0, little_endian: bool = False) -> KeygenParams:
    """One VALID (E', 3-isogeny onto y^2 = x^3 + 5) pair over the base field of `curve` (not necessarily pasta_curves' choice)."""
    p = CURVES[curve].base
    inv3 = pow(3, -1, p)
    roots = cube_roots(540, p)
    assert len(roots) == 3, "540 is not a cube in this field"
    x0 = roots[root_index]
    assert pow(x0, 3, p) == 540
    a = (-10 * inv3 * x0 * x0) % p
    b = 1265
    t = (-2 * inv3 * x0 * x0) % p
    u = 20
    i9, i27 = pow(9, -1, p), pow(27, -1, p)
    x_num = [i9, -2 * x0 * i9, (x0 * x0 + t) * i9, (u - t * x0) * i9]
    x_den = [-2 * x0, x0 * x0]
    y_num = [i27, -3 * x0 * i27, (3 * x0 * x0 - t) * i27, (-x0 ** 3 + t * x0 - 2 * u) * i27]
    y_den = [-3 * x0, 3 * x0 * x0, -x0 ** 3]
    z = find_z(p, a, b)
    dst = b"from_uniform_bytes-" + curve.encode() + b"_XMD:BLAKE2b_SSWU_RO_"
    return KeygenParams(p, a, b, z, [c % p for c in x_num + x_den + y_num + y_den], dst, little_endian)
