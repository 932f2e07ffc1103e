"""Host build of the kernels' field / group-law formulas (reef_amd/csrc/field.h, ec.h compiled
with g++ and REEF_BOUNDS) against the oracle.  Runs without a GPU: it checks the 29-bit limb
arithmetic bit-exactly and machine-checks every value-bound comment (a violated bound aborts
the process).  The GPU parity tests check the same formulas as compiled for gfx950."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle.pasta_oracle import CURVES, SplitMix64, uniform_scalar

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "reef_amd", "csrc")
SO = os.path.join(ROOT, "reef_amd", "_lib", "libreef_hostcheck.so")
CID = {"pallas": 0, "vesta": 1}


@pytest.fixture(scope="module")
def host():
    src = os.path.join(CSRC, "tools", "host_check.cpp")
    deps = [src] + [os.path.join(CSRC, f) for f in ("field.h", "ec.h", "field_consts.h", "glv_host.h", "glv_consts.h", "host_combine.h")]
    if not os.path.exists(SO) or os.path.getmtime(SO) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-DREEF_BOUNDS", "-shared", "-fPIC", src, "-o", SO])
    lib = ctypes.CDLL(SO)
    vp = ctypes.c_void_p
    lib.host_field_op.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_size_t]
    lib.host_ec_op.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_size_t]
    lib.host_accumulate.argtypes = [ctypes.c_int, vp, vp, ctypes.c_size_t, vp, vp, vp]
    lib.host_pack_roundtrip.argtypes = [vp, vp]
    lib.host_window_combine_check.argtypes = [ctypes.c_int, vp, ctypes.c_uint, ctypes.c_uint, vp, vp, vp]
    lib.host_glv_split.argtypes = [ctypes.c_int, vp, vp]
    lib.host_glv_phi.argtypes = [ctypes.c_int, vp, vp]
    return lib


def limbs(v):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def test_consts_generated_file_is_current():
    out = subprocess.check_output(["python", os.path.join(ROOT, "tools", "gen_field_consts.py")]).decode()
    assert out == open(os.path.join(CSRC, "field_consts.h")).read()


def test_pack_roundtrip(host):
    rng = SplitMix64(3)
    for v in [0, 1, (1 << 256) - 1, (1 << 255), (1 << 29) - 1, 1 << 29, (1 << 232)] + [rng.next256() for _ in range(200)]:
        a = np.array(limbs(v), dtype=np.uint64)
        o = np.zeros(4, dtype=np.uint64)
        host.host_pack_roundtrip(a.ctypes.data, o.ctypes.data)
        assert (a == o).all(), hex(v)


@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_field_ops(name, host, cref):
    f = CID[name]
    m = CURVES[name].base
    rng = SplitMix64(1234 + f)
    vals = [0, 1, 2, m - 1, m - 2, (1 << 254), (1 << 254) - 1, (1 << 128) - 1, 0xFFFFFFFF, 1 << 32, (1 << 29) - 1, 1 << 29,
            m - (1 << 29), (1 << 232), (1 << 232) - 1]
    vals += [uniform_scalar(rng, m) for _ in range(500)]
    n = len(vals)
    a = np.array([limbs(v) for v in vals], dtype=np.uint64)
    b = np.array([limbs(v) for v in reversed(vals)], dtype=np.uint64)
    out = np.zeros_like(a)
    for op, name_c, unary in ((0, "fmul", False), (1, "fadd", False), (2, "fsub", False), (3, "finv", True),
                              (4, "to_mont", True), (5, "from_mont", True)):
        host.host_field_op(f, op, a.ctypes.data, b.ctypes.data, out.ctypes.data, n)
        for i in range(n):
            exp = cref.field_op(name_c, f, a[i].copy()) if unary else cref.field_op(name_c, f, a[i].copy(), b[i].copy())
            assert (out[i] == exp).all(), (name_c, i, hex(vals[i]))
    rinv = pow(1 << 256, -1, m)
    host.host_field_op(f, 6, a.ctypes.data, b.ctypes.data, out.ctypes.data, n)
    for i in range(n):
        assert cref.limbs_to_int(out[i]) == (-vals[i]) % m
    host.host_field_op(f, 7, a.ctypes.data, b.ctypes.data, out.ctypes.data, n)
    for i in range(n):
        assert cref.limbs_to_int(out[i]) == vals[i] * vals[i] * rinv % m


@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_group_law(name, host, cref):
    cid = CID[name]
    C = CURVES[name]
    n = 48
    P = cref.gen_bases_ap(cid, 3, 5, n)
    Qp = cref.gen_bases_ap(cid, 1000, 9, n)
    Qp[0] = P[0]                                                            # doubling inside the add
    Qp[1] = np.frombuffer(C.affine_to_bytes(C.neg(C.affine_from_bytes(P[1].tobytes()))), dtype=np.uint64)  # P + (-P)
    Qp[2] = 0
    P[3] = 0
    P[4] = 0
    Qp[4] = 0
    rng = SplitMix64(5)
    ks = [uniform_scalar(rng, C.order) for _ in range(n)]
    ks[0], ks[1], ks[2] = 0, 1, C.order - 1
    k = np.array([limbs(v) for v in ks], dtype=np.uint64)
    out = np.zeros((n, 12), dtype=np.uint64)
    pts = [C.affine_from_bytes(P[i].tobytes()) for i in range(n)]
    qts = [C.affine_from_bytes(Qp[i].tobytes()) for i in range(n)]
    for op in (0, 1, 2, 3):
        host.host_ec_op(cid, op, P.ctypes.data, Qp.ctypes.data, k.ctypes.data, out.ctypes.data, n)
        comp = cref.compress(cid, out)
        for i in range(n):
            exp = {0: lambda: C.add(pts[i], qts[i]), 1: lambda: C.add(pts[i], qts[i]), 2: lambda: C.add(pts[i], pts[i]),
                   3: lambda: C.mul(ks[i], pts[i])}[op]()
            assert comp[32 * i:32 * i + 32] == C.compress(exp), (op, i)


@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_accumulate_chain_bounds(name, host, cref):
    """Long madd chains with sign flips, repeated points (P + P), cancellations (P + (-P)) and
    identity entries: steady-state bounds hold (else abort) and the sum matches the oracle,
    through all three output encodings."""
    cid = CID[name]
    C = CURVES[name]
    n = 600
    pts = cref.gen_bases_ap(cid, 17, 3, n)
    neg = (np.arange(n) % 3 == 1).astype(np.uint8)
    pts[5] = pts[4]; neg[5] = neg[4] = 0      # running sum meets... not necessarily equal, but repeated base
    pts[10] = 0                                # identity entry
    pts[20] = pts[19]; neg[19] = 0; neg[20] = 1   # immediate cancellation of the last addend
    jac = np.zeros(12, dtype=np.uint64)
    aff = np.zeros(8, dtype=np.uint64)
    comp = np.zeros(4, dtype=np.uint64)
    host.host_accumulate(cid, pts.ctypes.data, neg.ctypes.data, n, jac.ctypes.data, aff.ctypes.data, comp.ctypes.data)
    acc = None
    for i in range(n):
        p = C.affine_from_bytes(pts[i].tobytes())
        acc = C.add(acc, C.neg(p) if neg[i] else p)
    assert cref.compress(cid, jac) == C.compress(acc)
    assert aff.tobytes() == C.affine_to_bytes(acc)
    assert comp.tobytes() == C.compress(acc)
    # all-cancelling chain and single-point doubling chain
    two = np.stack([pts[7], pts[7]])
    host.host_accumulate(cid, two.ctypes.data, np.array([0, 1], dtype=np.uint8).ctypes.data, 2, jac.ctypes.data, aff.ctypes.data, comp.ctypes.data)
    assert cref.compress(cid, jac) == bytes(32) and comp.tobytes() == bytes(32) and (jac[8:] == 0).all()
    host.host_accumulate(cid, two.ctypes.data, np.array([0, 0], dtype=np.uint8).ctypes.data, 2, jac.ctypes.data, aff.ctypes.data, comp.ctypes.data)
    p7 = C.affine_from_bytes(pts[7].tobytes())
    assert cref.compress(cid, jac) == C.compress(C.add(p7, p7))


@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_window_combine_on_the_host(name, host, cref):
    """host_combine.h -- the window combine of a plain key on four 64-bit limbs, what the library runs on a host core since round 6 -- against
    ec.h's own chain on the same window sums and against the big-integer oracle: sum_g 2^(c*g) * S_g for the plans the engine picks (c = 8 ... 16,
    G = ceil(255 / c)), with empty windows, an empty top window, a window that equals the running sum (the doubling branch of the addition), one that
    cancels it (the identity in mid-chain) and an all-empty input.  The window sums reach it in the engine's memory form, un-normalised limbs included."""
    cid = CID[name]
    C = CURVES[name]

    def to_row(pt):
        return np.frombuffer(C.affine_to_bytes(pt), dtype=np.uint64)

    def run(sums, c):                                  # sums: per window a pair of affine points (None = identity) whose sum is the window's
        G = len(sums)
        pts = np.zeros((2 * G, 8), dtype=np.uint64)
        for g, (a, b) in enumerate(sums):
            pts[2 * g], pts[2 * g + 1] = to_row(a), to_row(b)
        new, old, us = np.zeros(12, dtype=np.uint64), np.zeros(12, dtype=np.uint64), np.zeros(2)
        host.host_window_combine_check(cid, pts.ctypes.data, G, c, new.ctypes.data, old.ctypes.data, us.ctypes.data)
        want = None
        for g in reversed(range(G)):
            for _ in range(c):
                want = C.add(want, want)
            want = C.add(want, C.add(sums[g][0], sums[g][1]))
        assert cref.compress(cid, new) == C.compress(want), ("host_combine.h", c, G)
        assert cref.compress(cid, old) == C.compress(want), ("ec.h", c, G)
        if want is None:
            assert not new.any()                       # the identity is (0, 0, 0) at the ABI
        return us

    base = cref.gen_bases_ap(cid, 41, 3, 80)
    P = [C.affine_from_bytes(base[i].tobytes()) for i in range(80)]
    for c in (8, 11, 13, 16):
        G = (255 + c - 1) // c
        sums = [(P[2 * g], P[2 * g + 1]) for g in range(G)]
        sums[1] = (None, None)                         # an empty window
        sums[2] = (P[5], None)                         # a single point (ZZ = 1)
        sums[3] = (None, P[6])
        us = run(sums, c)
        sums[G - 1] = (None, None)                     # nothing in the top window: the chain starts from the identity
        sums[G - 2] = (P[9], C.neg(P[9]))              # a pair that cancels inside a window
        run(sums, c)
    c = 8
    top = C.add(P[0], P[1])
    shifted = top
    for _ in range(c):
        shifted = C.add(shifted, shifted)
    run([(P[7], None), (shifted, None), (P[0], P[1])], c)            # window 1 equals 2^c * (window 2): the addition's doubling branch
    run([(P[7], None), (C.neg(shifted), None), (P[0], P[1])], c)     # ... and its negative: the identity in mid-chain, the chain goes on from P[7]
    run([(None, None)] * 4, c)
    rng = SplitMix64(77 + cid)                         # random plans: any window width the engine may pick, random empty windows and single points
    for _ in range(24):
        c = 5 + rng.next() % 13
        G = (255 + c - 1) // c
        sums = []
        for g in range(G):
            kind = rng.next() % 5
            a, b = P[rng.next() % 80], P[rng.next() % 80]
            sums.append([(a, b), (a, None), (None, b), (None, None), (a, a)][kind])      # (a, a): the mixed addition's own doubling branch inside a window
        run(sums, c)
    print(f"{name}: window combine c = 16, G = 16: {us[0]:.0f} us on four 64-bit limbs, {us[1]:.0f} us through ec.h (g++ -O2 with bound tracking)")


@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_glv_split_and_endomorphism(name, host, cref):
    """The generator fold halves its doubling chain with the curve endomorphism: phi(x, y) = (beta*x, y) must be
    multiplication by lambda, and glv_split must return k = k1 + k2*lambda (mod r) with both parts below 2^128."""
    import re
    C = CURVES[name]
    cid = CID[name]
    r = C.order
    hdr = open(os.path.join(CSRC, "glv_consts.h")).read()
    blk = hdr[hdr.index("struct GLV<%d>" % cid):]
    words = re.search(r"LAMBDA\[4\] = \{([^}]*)\}", blk).group(1)
    lam = sum(int(w.strip().rstrip("ull"), 16) << (64 * i) for i, w in enumerate(words.split(",")))
    assert (lam * lam + lam + 1) % r == 0
    # phi(P) = lambda * P on a few points, through the shared field code
    pts = cref.gen_bases_ap(cid, 12345, 678, 4)
    for i in range(4):
        out = np.zeros(8, dtype=np.uint64)
        host.host_glv_phi(cid, pts[i].ctypes.data, out.ctypes.data)
        P = C.affine_from_bytes(pts[i].tobytes())
        assert C.affine_from_bytes(out.tobytes()) == C.mul(lam, P)
    rng = SplitMix64(42)
    ks = [0, 1, 2, r - 1, r - 2, lam, (lam * lam) % r, 1 << 128, (1 << 254) - 1] + [uniform_scalar(rng, r) for _ in range(300)]
    for k in ks:
        kw = np.array([(k >> (32 * i)) & 0xFFFFFFFF for i in range(8)], dtype=np.uint32)
        out = np.zeros(12, dtype=np.uint32)
        assert host.host_glv_split(cid, kw.ctypes.data, out.ctypes.data) == 1
        k1 = sum(int(out[i]) << (32 * i) for i in range(5)) * (-1 if out[10] else 1)
        k2 = sum(int(out[5 + i]) << (32 * i) for i in range(5)) * (-1 if out[11] else 1)
        assert (k1 + k2 * lam - k) % r == 0, hex(k)
        assert abs(k1) < 1 << 128 and abs(k2) < 1 << 128, hex(k)


def test_glv_consts_generated_file_is_current():
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_glv_consts.py")], capture_output=True, text=True, check=True).stdout
    assert gen == open(os.path.join(CSRC, "glv_consts.h")).read(), "run tools/gen_glv_consts.py > reef_amd/csrc/glv_consts.h"

