"""Big-integer oracle for the Pasta-curve MSM hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
anything under oracle/.  The product path (reef_amd + libreef_msm.so) never does.

PARITY UNPINNED BY THE REFERENCE: the arithmetic this restates does not live in
/root/reference (it is in the un-vendored crates nova-snark @ sga001/Nova (no rev),
fil_pasta_curves 0.5.2 and pasta-msm 0.1.x, see Cargo.toml:12,14) and Reef's own
tests hold no MSM known-answer vector (every prover test is a prove->verify
round-trip with OsRng blinds, src/backend/commitment.rs:152,348,359,419,421).
This oracle therefore follows the *published definition* of the curves:

  Pallas : y^2 = x^3 + 5 over Fp, prime order q, generator (-1, 2)
  Vesta  : y^2 = x^3 + 5 over Fq, prime order p, generator (-1, 2)

and is pinned by (i) the scalar modulus Reef itself hard-codes
(src/backend/r1cs_helper.rs:37-38 == q below), (ii) group-law invariants
(q*G = O on Pallas, p*G = O on Vesta, on-curve checks), (iii) agreement with an
independent C restatement (oracle/pasta_ref.c) and (iv) the anchor vectors of
SURVEY.md section 8c.  Every function names the reference call site whose
semantics it restates.

Pure Python ints; written for clarity, not speed.
"""
from __future__ import annotations

import hashlib
from typing import Iterable, List, Optional, Sequence, Tuple

# ---------------------------------------------------------------------------
# Moduli.  q is the CirC field modulus at src/backend/r1cs_helper.rs:37-38.
# ---------------------------------------------------------------------------
P = 0x40000000000000000000000000000000224698FC094CF91B992D30ED00000001  # Pallas base field
Q = 0x40000000000000000000000000000000224698FC0994A8DD8C46EB2100000001  # Pallas scalar field
R256 = 1 << 256
B_COEFF = 5

Affine = Optional[Tuple[int, int]]  # None == point at infinity


class Curve:
    """One of the two Pasta curves.

    `base` is the coordinate field modulus, `order` the scalar field modulus.
    G1 = pallas::Point, G2 = vesta::Point at src/backend/framework.rs:1-2.
    """

    def __init__(self, name: str, base: int, order: int):
        self.name = name
        self.base = base
        self.order = order
        self.gen: Affine = (base - 1, 2)

    # -- field helpers ------------------------------------------------------
    def inv(self, a: int) -> int:
        return pow(a, -1, self.base)

    def is_on_curve(self, pt: Affine) -> bool:
        if pt is None:
            return True
        x, y = pt
        return (y * y - x * x * x - B_COEFF) % self.base == 0

    # -- group law (affine, textbook) --------------------------------------
    def neg(self, pt: Affine) -> Affine:
        if pt is None:
            return None
        return (pt[0], (-pt[1]) % self.base)

    def add(self, a: Affine, b: Affine) -> Affine:
        if a is None:
            return b
        if b is None:
            return a
        p = self.base
        x1, y1 = a
        x2, y2 = b
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = 3 * x1 * x1 * self.inv(2 * y1) % p
        else:
            lam = (y2 - y1) * self.inv(x2 - x1) % p
        x3 = (lam * lam - x1 - x2) % p
        y3 = (lam * (x1 - x3) - y1) % p
        return (x3, y3)

    def mul(self, k: int, pt: Affine) -> Affine:
        """k*pt by plain double-and-add (k reduced mod the group order)."""
        k %= self.order
        acc: Affine = None
        addend = pt
        while k:
            if k & 1:
                acc = self.add(acc, addend)
            addend = self.add(addend, addend)
            k >>= 1
        return acc

    # -- MSM: restates Group::vartime_multiscalar_mul (nova-snark provider/pasta.rs,
    #    reached from src/backend/commitment.rs:187,350,361,371,383,422,430 and
    #    src/backend/framework.rs:668,695) by its definition sum_i s_i * P_i. ----
    def msm_naive(self, scalars: Sequence[int], bases: Sequence[Affine]) -> Affine:
        assert len(scalars) == len(bases)
        acc: Affine = None
        for s, b in zip(scalars, bases):
            acc = self.add(acc, self.mul(s, b))
        return acc

    def msm(self, scalars: Sequence[int], bases: Sequence[Affine], c: int = 8) -> Affine:
        """Bucket-method MSM (unsigned c-bit windows); same result as msm_naive.

        Restates the bucket method of halo2-style `cpu_best_multiexp` that nova-snark
        falls back to for n < 128, with a fixed window.  Jacobian inside for speed.
        """
        assert len(scalars) == len(bases)
        nwin = (255 + c - 1) // c
        total = JAC_INF
        for w in reversed(range(nwin)):
            for _ in range(c):
                total = self.jdbl(total)
            buckets = [JAC_INF] * ((1 << c) - 1)
            for s, b in zip(scalars, bases):
                if b is None:
                    continue
                d = ((s % self.order) >> (w * c)) & ((1 << c) - 1)
                if d:
                    buckets[d - 1] = self.jadd_mixed(buckets[d - 1], b)
            run = JAC_INF
            acc = JAC_INF
            for bk in reversed(buckets):
                run = self.jadd(run, bk)
                acc = self.jadd(acc, run)
            total = self.jadd(total, acc)
        return self.to_affine(total)

    # -- Jacobian helpers (x, y, z) with z == 0 <=> infinity -------------------
    def jdbl(self, a):
        p = self.base
        x, y, z = a
        if z == 0 or y == 0:
            return JAC_INF
        s = 4 * x * y * y % p
        m = 3 * x * x % p
        x3 = (m * m - 2 * s) % p
        y3 = (m * (s - x3) - 8 * y * y * y * y) % p
        z3 = 2 * y * z % p
        return (x3, y3, z3)

    def jadd(self, a, b):
        p = self.base
        if a[2] == 0:
            return b
        if b[2] == 0:
            return a
        x1, y1, z1 = a
        x2, y2, z2 = b
        z1z1 = z1 * z1 % p
        z2z2 = z2 * z2 % p
        u1 = x1 * z2z2 % p
        u2 = x2 * z1z1 % p
        s1 = y1 * z2 * z2z2 % p
        s2 = y2 * z1 * z1z1 % p
        if u1 == u2:
            if s1 == s2:
                return self.jdbl(a)
            return JAC_INF
        h = (u2 - u1) % p
        r = (s2 - s1) % p
        hh = h * h % p
        hhh = h * hh % p
        v = u1 * hh % p
        x3 = (r * r - hhh - 2 * v) % p
        y3 = (r * (v - x3) - s1 * hhh) % p
        z3 = z1 * z2 * h % p
        return (x3, y3, z3)

    def jadd_mixed(self, a, b: Affine):
        if b is None:
            return a
        return self.jadd(a, (b[0], b[1], 1))

    def to_affine(self, a) -> Affine:
        x, y, z = a
        if z == 0:
            return None
        zi = self.inv(z)
        zi2 = zi * zi % self.base
        return (x * zi2 % self.base, y * zi2 * zi % self.base)

    # -- encodings ------------------------------------------------------------
    def to_mont(self, a: int) -> int:
        """Coordinate -> Montgomery form (R = 2^256), as fil_pasta_curves stores Fp/Fq."""
        return a * R256 % self.base

    def from_mont(self, a: int) -> int:
        return a * pow(R256, -1, self.base) % self.base

    def scalar_to_mont(self, s: int) -> int:
        return s * R256 % self.order

    def scalar_from_mont(self, s: int) -> int:
        return s * pow(R256, -1, self.order) % self.order

    def compress(self, pt: Affine) -> bytes:
        """pasta_curves GroupEncoding::to_bytes (used through Commitment::compress at
        src/backend/commitment.rs:195,351,365,425,427,431): little-endian canonical x,
        y-parity in bit 255; identity = 32 zero bytes."""
        if pt is None:
            return bytes(32)
        x, y = pt
        b = bytearray(x.to_bytes(32, "little"))
        b[31] |= (y & 1) << 7
        return bytes(b)

    def decompress(self, data: bytes) -> Affine:
        assert len(data) == 32
        if data == bytes(32):
            return None
        sign = data[31] >> 7
        x = int.from_bytes(data, "little") & ((1 << 255) - 1)
        assert x < self.base
        y2 = (x * x * x + B_COEFF) % self.base
        y = sqrt_mod(y2, self.base)
        assert y is not None, "x not on curve"
        if (y & 1) != sign:
            y = self.base - y
        return (x, y)

    # -- raw little-endian layouts handed across the C ABI ---------------------
    def affine_to_bytes(self, pt: Affine) -> bytes:
        """64-byte repr(C) EpAffine/EqAffine: x,y as 4xu64 LE Montgomery limbs;
        identity = (0, 0)."""
        if pt is None:
            return bytes(64)
        return self.to_mont(pt[0]).to_bytes(32, "little") + self.to_mont(pt[1]).to_bytes(32, "little")

    def affine_from_bytes(self, data: bytes) -> Affine:
        x = int.from_bytes(data[:32], "little")
        y = int.from_bytes(data[32:64], "little")
        if x == 0 and y == 0:
            return None
        return (self.from_mont(x), self.from_mont(y))

    def jacobian_from_bytes(self, data: bytes) -> Affine:
        """96-byte repr(C) Ep/Eq (x, y, z Montgomery) -> canonical affine."""
        x = self.from_mont(int.from_bytes(data[:32], "little"))
        y = self.from_mont(int.from_bytes(data[32:64], "little"))
        z = self.from_mont(int.from_bytes(data[64:96], "little"))
        return self.to_affine((x, y, z))

    def scalar_to_bytes(self, s: int, mont: bool = True) -> bytes:
        s %= self.order
        return (self.scalar_to_mont(s) if mont else s).to_bytes(32, "little")


JAC_INF = (0, 1, 0)

PALLAS = Curve("pallas", P, Q)
VESTA = Curve("vesta", Q, P)
CURVES = {"pallas": PALLAS, "vesta": VESTA}


def sqrt_mod(a: int, p: int) -> Optional[int]:
    """Tonelli-Shanks (both Pasta fields have 2-adicity 32)."""
    a %= p
    if a == 0:
        return 0
    if pow(a, (p - 1) // 2, p) != 1:
        return None
    s, t = 0, p - 1
    while t % 2 == 0:
        s += 1
        t //= 2
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, tt, r = s, pow(z, t, p), pow(a, t, p), pow(a, (t + 1) // 2, p)
    while tt != 1:
        i, x = 0, tt
        while x != 1:
            x = x * x % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c = i, b * b % p
        tt = tt * c % p
        r = r * b % p
    return r


# ---------------------------------------------------------------------------
# Deterministic test-vector generators (shared by gen_golden.py and tests)
# ---------------------------------------------------------------------------
class SplitMix64:
    """Same generator as reef_amd/csrc/reef_rng.h so that C, HIP and Python agree."""

    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def next256(self) -> int:
        return self.next() | (self.next() << 64) | (self.next() << 128) | (self.next() << 192)


def uniform_scalar(rng: SplitMix64, order: int) -> int:
    """Uniform-ish scalar: 256 random bits with the top bit cleared, reduced once."""
    v = rng.next256() & ((1 << 255) - 1)
    return v - order if v >= order else v


def ap_bases(curve: Curve, k0: int, d: int, n: int) -> List[Affine]:
    """Bases in arithmetic progression B_i = (k0 + i*d) * G: discrete logs are known,
    so  MSM(s, B) == (sum_i s_i*(k0+i*d) mod order) * G  gives a size-independent check."""
    out: List[Affine] = []
    cur = curve.mul(k0, curve.gen)
    step = curve.mul(d, curve.gen)
    for _ in range(n):
        out.append(cur)
        cur = curve.add(cur, step)
    return out


def msm_via_dlog(curve: Curve, scalars: Iterable[int], k0: int, d: int) -> Affine:
    acc = 0
    for i, s in enumerate(scalars):
        acc = (acc + s * (k0 + i * d)) % curve.order
    return curve.mul(acc, curve.gen)


def sha_hex(data: bytes) -> str:
    return hashlib.sha256(data).hexdigest()
