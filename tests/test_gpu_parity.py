"""Parity of the HIP path (through the C ABI) against the oracle and the golden vectors.

Bit-exact on the canonical encodings (affine Montgomery limbs / 32-byte compressed points):
Jacobian representatives are not unique, so outputs are normalised before comparison.
"""
import hashlib
import threading

import numpy as np
import pytest

from oracle.pasta_oracle import CURVES, SplitMix64, msm_via_dlog, uniform_scalar

pytestmark = pytest.mark.gpu
CID = {"pallas": 0, "vesta": 1}


def limbs(v):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def explicit_arrays(case):
    n = len(case["bases_hex"])
    if n == 0:
        return np.zeros((0, 8), np.uint64), np.zeros((0, 4), np.uint64), np.zeros((0, 4), np.uint64)
    bases = np.frombuffer(b"".join(bytes.fromhex(h) for h in case["bases_hex"]), dtype=np.uint64).reshape(n, 8).copy()
    sm = np.frombuffer(b"".join(bytes.fromhex(h) for h in case["scalars_mont_hex"]), dtype=np.uint64).reshape(n, 4).copy()
    sc = np.frombuffer(b"".join(bytes.fromhex(h) for h in case["scalars_canon_hex"]), dtype=np.uint64).reshape(n, 4).copy()
    return bases, sm, sc


# ------------------------------------------------------------------ field / group law ----
@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_field_ops(name, gpu_lib, cref):
    f = CID[name]
    m = CURVES[name].base
    rng = SplitMix64(1234 + f)
    vals = [0, 1, 2, m - 1, m - 2, (1 << 255) % m, (1 << 254), (1 << 128) - 1, 0xFFFFFFFF, 1 << 32]
    vals += [uniform_scalar(rng, m) for _ in range(246)]
    n = len(vals)
    a = np.array([limbs(v) for v in vals], dtype=np.uint64)
    b = np.array([limbs(v) for v in reversed(vals)], dtype=np.uint64)
    out = np.zeros_like(a)
    for op, name_c, unary in ((0, "fmul", False), (1, "fadd", False), (2, "fsub", False), (3, "finv", True),
                              (4, "to_mont", True), (5, "from_mont", True)):
        assert gpu_lib.reef_test_field_op(f, op, a.ctypes.data, b.ctypes.data, out.ctypes.data, n) == 0
        for i in range(n):
            exp = cref.field_op(name_c, f, a[i].copy()) if unary else cref.field_op(name_c, f, a[i].copy(), b[i].copy())
            assert (out[i] == exp).all(), (name_c, i, hex(vals[i]))
    # neg and sqr against big ints
    assert gpu_lib.reef_test_field_op(f, 6, a.ctypes.data, b.ctypes.data, out.ctypes.data, n) == 0
    for i in range(n):
        assert cref.limbs_to_int(out[i]) == (-vals[i]) % m
    assert gpu_lib.reef_test_field_op(f, 7, a.ctypes.data, b.ctypes.data, out.ctypes.data, n) == 0
    rinv = pow(1 << 256, -1, m)
    for i in range(n):
        assert cref.limbs_to_int(out[i]) == vals[i] * vals[i] * rinv % m


@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_group_law(name, gpu_lib, cref):
    from reef_amd import msm
    cid = CID[name]
    C = CURVES[name]
    n = 150                                 # three workgroups of the four-wave kernels, the last one ragged
    P = cref.gen_bases_ap(cid, 3, 5, n)
    Qp = cref.gen_bases_ap(cid, 1000, 9, n)
    # special cases: P == Q (doubling), P == -Q, identity operands
    Qp[0] = P[0]
    Qp[1] = np.frombuffer(C.affine_to_bytes(C.neg(C.affine_from_bytes(P[1].tobytes()))), dtype=np.uint64)
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
    # 4-6, 11-19: the same operations shared by the four waves of a workgroup (the tail kernels' form), alone and in every
    # back-to-back order -- hipcc's load-store vectorizer miscompiled doubling-after-doubling (ec_coop.h, tools/bisect/)
    A, D = (lambda u, v: C.add(u, v)), (lambda u: C.add(u, u))
    expect = {0: lambda p, q, k_: C.add(p, q), 1: lambda p, q, k_: C.add(p, q), 2: lambda p, q, k_: D(p), 3: lambda p, q, k_: C.mul(k_, p),
              4: lambda p, q, k_: A(p, q), 5: lambda p, q, k_: D(p), 6: lambda p, q, k_: A(D(D(A(p, q))), p),
              11: lambda p, q, k_: D(A(p, q)), 12: lambda p, q, k_: A(A(p, q), p), 13: lambda p, q, k_: D(D(A(p, q))),
              14: lambda p, q, k_: A(D(p), q), 15: lambda p, q, k_: A(D(D(p)), q), 16: lambda p, q, k_: D(D(p)),
              17: lambda p, q, k_: D(D(p)), 18: lambda p, q, k_: D(D(p)), 19: lambda p, q, k_: D(D(p)),
              20: lambda p, q, k_: D(D(p)), 21: lambda p, q, k_: A(p, q), 22: lambda p, q, k_: C.mul(7, A(p, q))}
    for op in sorted(expect):
        assert gpu_lib.reef_test_ec_op(cid, op, P.ctypes.data, Qp.ctypes.data, k.ctypes.data, out.ctypes.data, n) == 0
        comp = cref.compress(cid, out)
        gcomp = msm.compress(cid, out)
        assert comp == gcomp  # K4 normalise/compress on the GPU == oracle
        for i in range(n):
            assert comp[32 * i:32 * i + 32] == C.compress(expect[op](pts[i], qts[i], ks[i])), (op, i)


# ---------------------------------------------------------------------------- K1: MSM ----
def test_golden_explicit_stateless(golden, gpu_lib):
    """pasta-msm drop-in symbols, both scalar conventions (is_mont true / false)."""
    from reef_amd import msm
    for case in golden["explicit"]:
        bases, sm, sc = explicit_arrays(case)
        for scal, mont in ((sm, True), (sc, False)):
            r = msm.mult_pippenger(case["curve"], bases, scal, is_mont=mont)
            assert msm.compress(case["curve"], r).hex() == case["expect_compressed"], (case["label"], mont)


@pytest.mark.parametrize("groups", [0, 1, 4])
def test_golden_explicit_handle(golden, gpu_lib, groups):
    from reef_amd import msm
    for case in golden["explicit"]:
        bases, sm, sc = explicit_arrays(case)
        if bases.shape[0] == 0:
            continue
        with msm.MsmContext(case["curve"], bases, window_bits=5, bucket_groups=groups) as ctx:
            r = ctx.msm(sm)
            assert msm.compress(case["curve"], r).hex() == case["expect_compressed"], (case["label"], groups)
            r = ctx.msm(sc, is_mont=False)
            assert msm.compress(case["curve"], r).hex() == case["expect_compressed"], (case["label"], groups)


def test_golden_seeded(golden, gpu_lib):
    """Sizes {1,2,3,127,128,129,1000,4096}, uniform and witness-like scalars; inputs generated on
    the GPU must be byte-identical to the oracle's (sha256 pinned in the fixture)."""
    from reef_amd import msm
    for case in golden["seeded"]:
        n = case["n"]
        bases = msm.gen_bases(case["curve"], case["k0"], case["d"], n)
        sc = msm.gen_scalars(case["curve"], case["seed"], n, kind=case["kind"])
        assert hashlib.sha256(bases.tobytes() + sc.tobytes()).hexdigest() == case["input_sha256"], (case["curve"], n)
        r = msm.mult_pippenger(case["curve"], bases, sc)
        assert msm.compress(case["curve"], r).hex() == case["expect_compressed"], (case["curve"], n, case["kind"])


@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_device_resident_blinding_generator(name, gpu_lib, cref):
    """CE::commit / HyraxPC::commit with every buffer on the device (commitment.rs:187, :349-351): the table of h follows a
    device-resident h WITHOUT a host round trip (k_h_refresh) -- rewritten in place, the identity, and mixed with calls
    that pass h from the host."""
    from reef_amd import msm
    from reef_amd import _ffi
    cid = msm.curve_id(name)
    rows, row_len = 37, 700
    bases = cref.gen_bases_ap(cid, 0xD0C, 3, row_len)
    sc = cref.gen_scalars(cid, 31, rows * row_len)
    bl = cref.gen_scalars(cid, 32, rows)
    hs = [cref.gen_bases_ap(cid, 0xB11D + k, 1 + k, 1)[0].copy() for k in range(3)]
    hs.append(np.zeros_like(hs[0]))                      # the identity (0, 0): the blind term vanishes
    want = [cref.compress(cid, cref.row_msm(cid, bases, sc, rows, row_len, h=h, blinds=bl, threads=4)) for h in hs]
    d_sc, d_bl, d_h = msm.DeviceBuffer.from_host(sc), msm.DeviceBuffer.from_host(bl), msm.DeviceBuffer.from_host(hs[0])
    out = msm.DeviceBuffer(96 * rows)

    def put(h):
        h = np.ascontiguousarray(h)
        msm.check(_ffi.load().reef_memcpy(d_h.ptr, h.ctypes.data, h.nbytes, msm.REEF_DEVICE, msm.REEF_HOST))

    def run_device(ctx):
        ctx.msm_rows(d_sc, rows, row_len, blinds=d_bl, h=d_h, out=out)
        ctx.sync()
        return msm.compress(cid, out.to_host((rows, 12)))

    with msm.MsmContext(name, bases) as ctx:
        for rep in range(2):                             # the second pass meets a table built for the last h of the first
            for k in (0, 0, 1, 3, 2, 2, 0):
                put(hs[k])
                assert run_device(ctx) == want[k], (rep, k)
            # h from the host in between: each path must see that the other one rebuilt the table
            assert msm.compress(cid, ctx.msm_rows(sc, rows, row_len, blinds=bl, h=hs[1])) == want[1]
            put(hs[0])
            assert run_device(ctx) == want[0]
            assert msm.compress(cid, ctx.msm_rows(sc, rows, row_len, blinds=bl, h=hs[1])) == want[1]
            put(hs[1])
            assert run_device(ctx) == want[1]
        # blinds on the edges of the nibble decomposition (canonical form): 0, 1, r - 1, every nibble 1 / 8 / 15
        r_ = CURVES[name].order
        edge = [0, 1, r_ - 1, int("1" * 63, 16), int("8" * 63, 16) % r_, int("f" * 64, 16) % r_, 15 << 252 if (15 << 252) < r_ else 3 << 252]
        e_sc = cref.gen_scalars(cid, 35, len(edge) * row_len, mont=False)
        e_bl = np.array([[(b >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for b in edge], dtype=np.uint64)
        e_want = cref.compress(cid, cref.row_msm(cid, bases, e_sc, len(edge), row_len, h=hs[2], blinds=e_bl, mont=False, threads=2))
        d_esc, d_ebl = msm.DeviceBuffer.from_host(e_sc), msm.DeviceBuffer.from_host(e_bl)
        e_out = msm.DeviceBuffer(96 * len(edge))
        put(hs[2])
        ctx.msm_rows(d_esc, len(edge), row_len, is_mont=False, blinds=d_ebl, h=d_h, out=e_out)
        ctx.sync()
        assert msm.compress(cid, e_out.to_host((len(edge), 12))) == e_want
        assert msm.compress(cid, ctx.msm_rows(e_sc, len(edge), row_len, is_mont=False, blinds=e_bl, h=hs[2])) == e_want
        # one short commitment (the nibble-table path takes h's table as its second segment)
        v = cref.gen_scalars(cid, 33, 300)
        b1 = cref.gen_scalars(cid, 34, 1)
        d_v, d_b1 = msm.DeviceBuffer.from_host(v), msm.DeviceBuffer.from_host(b1)
        one = msm.DeviceBuffer(96)
        for k in (2, 0):
            put(hs[k])
            ctx.msm_rows(d_v, 1, 300, blinds=d_b1, h=d_h, out=one)
            ctx.sync()
            assert msm.compress(cid, one.to_host((1, 12))) == cref.compress(cid, cref.row_msm(cid, bases, v, 1, 300, h=hs[k], blinds=b1, threads=2))


@pytest.mark.parametrize("name", ["pallas", "vesta"])
@pytest.mark.parametrize("plan", [(0, 0), (7, 0), (7, 1), (9, 3), (12, 1), (13, 0), (16, 0), (4, 2)])
def test_plans_vs_c_oracle(name, plan, gpu_lib, cref):
    """Every (window, bucket-group) plan gives the oracle's answer; prefix MSMs (n < key length)."""
    from reef_amd import msm
    cid = CID[name]
    n = 3000
    bases = cref.gen_bases_ap(cid, 17, 3, n)
    with msm.MsmContext(cid, bases, window_bits=plan[0], bucket_groups=plan[1]) as ctx:
        for kind, m in ((0, n), (1, n), (0, 777), (2, 1500)):
            sc = cref.gen_scalars(cid, 99 + kind + m, m, kind=kind, small_bound=131)
            exp = cref.compress(cid, cref.msm_pippenger(cid, bases[:m].copy(), sc, threads=4))
            got = msm.compress(cid, ctx.msm(sc))
            assert got == exp, (plan, kind, m)


@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_skewed_and_degenerate_inputs(name, gpu_lib, cref):
    """Heavy buckets (all scalars equal), duplicate bases (P+P inside a bucket), cancelling pairs."""
    from reef_amd import msm
    cid = CID[name]
    C = CURVES[name]
    n = 5000
    bases = cref.gen_bases_ap(cid, 5, 1, n)
    with msm.MsmContext(cid, bases) as ctx:
        for val in (1, 2, C.order - 1, 0x8000, 0xFFFF):
            sc = np.tile(np.array(limbs(val), dtype=np.uint64), (n, 1))
            exp = cref.compress(cid, cref.msm_pippenger(cid, bases, sc, mont=False, threads=4))
            assert msm.compress(cid, ctx.msm(sc, is_mont=False)) == exp, hex(val)
    dup = np.tile(bases[7], (n, 1))
    dup[1::2] = np.frombuffer(C.affine_to_bytes(C.neg(C.affine_from_bytes(bases[7].tobytes()))), dtype=np.uint64)
    with msm.MsmContext(cid, dup) as ctx:
        sc = np.tile(np.array(limbs(3), dtype=np.uint64), (n, 1))
        assert msm.compress(cid, ctx.msm(sc, is_mont=False)) == bytes(32)  # everything cancels
        sc = cref.gen_scalars(cid, 1, n, kind=0)
        exp = cref.compress(cid, cref.msm_pippenger(cid, dup, sc, threads=4))
        assert msm.compress(cid, ctx.msm(sc)) == exp
    same = np.tile(bases[9], (64, 1))
    with msm.MsmContext(cid, same) as ctx:
        sc = np.tile(np.array(limbs(5), dtype=np.uint64), (64, 1))
        pt = C.mul(5 * 64, C.affine_from_bytes(bases[9].tobytes()))
        assert msm.compress(cid, ctx.msm(sc, is_mont=False)) == C.compress(pt)


@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_ragged_sizes_prefixes_and_holes(name, gpu_lib, cref):
    """Every size around the wave / workgroup / pasta-msm dispatch edges (n = 128 is where nova switches
    to the C symbol), through the stateless symbol and as prefixes of one resident key, with identity
    bases and zero scalars sprinkled in (affine (0,0) and scalar 0 must both vanish)."""
    from reef_amd import msm
    cid = CID[name]
    nmax = 4100
    bases = cref.gen_bases_ap(cid, 21, 4, nmax)
    sc = cref.gen_scalars(cid, 404, nmax, kind=1)
    bases[5] = 0                     # identity base with a non-zero scalar
    bases[130] = 0
    sc[6] = 0                        # zero scalar on a real base
    sc[131] = 0
    bases[300] = 0
    sc[300] = 0                      # both
    sizes = [1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 4095, 4096, 4097, nmax]
    want = {n: cref.compress(cid, cref.msm_pippenger(cid, bases[:n].copy(), sc[:n].copy(), threads=4)) for n in sizes}
    for n in sizes:
        assert msm.compress(cid, msm.mult_pippenger(cid, bases[:n].copy(), sc[:n].copy())) == want[n], n
    for groups in (0, 1):
        with msm.MsmContext(cid, bases, bucket_groups=groups) as ctx:
            for n in sizes:
                assert msm.compress(cid, ctx.msm(sc[:n].copy())) == want[n], (groups, n)
            assert msm.compress(cid, ctx.msm(np.zeros((0, 4), dtype=np.uint64))) == bytes(32)   # empty MSM


@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_stateless_symbol_recognises_a_returning_key(name, gpu_lib, cref):
    """The drop-in symbol retains nothing the caller can see, but a key that keeps coming back is served
    from a resident pre-shifted copy from its third call on (fingerprint of the uploaded bytes).  Same
    results on every call; a key edited in place (same pointer, same length) must not be mistaken
    for the old one; more keys than cache slots still work."""
    from reef_amd import msm
    cid = CID[name]
    n = 3000
    bases = cref.gen_bases_ap(cid, 1001, 7, n)
    for rep in range(5):                                          # 1st: plain, 2nd: plain + resident copy built, 3rd..: resident
        sc = cref.gen_scalars(cid, 50 + rep, n, kind=rep % 2)
        assert msm.compress(cid, msm.mult_pippenger(cid, bases, sc)) == cref.compress(cid, cref.msm_pippenger(cid, bases, sc, threads=4)), rep
    sc = cref.gen_scalars(cid, 99, n)
    bases[17] = cref.gen_bases_ap(cid, 5, 1, 1)[0]                # in-place edit of one point
    want = cref.compress(cid, cref.msm_pippenger(cid, bases, sc, threads=4))
    for rep in range(4):
        assert msm.compress(cid, msm.mult_pippenger(cid, bases, sc)) == want, rep
    keys = [cref.gen_bases_ap(cid, 2000 + 13 * k, 3, 1500) for k in range(9)]     # more keys than slots, revisited
    sck = cref.gen_scalars(cid, 7, 1500)
    wants = [cref.compress(cid, cref.msm_pippenger(cid, kb, sck, threads=4)) for kb in keys]
    for rnd in range(3):
        for kb, w in zip(keys, wants):
            assert msm.compress(cid, msm.mult_pippenger(cid, kb, sck)) == w, rnd


@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_small_resident_keys_take_the_nibble_table_path(name, gpu_lib, cref):
    """Keys of at most 1024 points with pre-shifted tables (bucket_groups = 1) are served from nibble tables (k_small_msm:
    no sort, no buckets): every size from one point (CE::commit, src/backend/commitment.rs:349-361,422,430) to the
    cap_prove sizes (commitment.rs:261-268), prefixes, both scalar conventions, edge scalars, identity and duplicate
    bases, and the commitment with a blind v*G + b*H as reef_msm_rows(rows = 1)."""
    from reef_amd import msm
    cid = CID[name]
    C = CURVES[name]
    rng = SplitMix64(2024 + cid)
    h = cref.gen_bases_ap(cid, 0xB11D, 1, 1)[0].copy()
    h2 = cref.gen_bases_ap(cid, 0xB11E, 1, 1)[0].copy()
    for n in (1, 2, 3, 8, 9, 63, 64, 65, 130, 600, 1024):
        bases = cref.gen_bases_ap(cid, 31 + n, 7, n)
        if n >= 8:
            bases[5] = 0                      # identity base
            bases[6] = bases[4]               # duplicate: equal table entries meet in the lane tree (P + P)
            bases[7] = np.frombuffer(C.affine_to_bytes(C.neg(C.affine_from_bytes(bases[4].tobytes()))), dtype=np.uint64)
        with msm.MsmContext(cid, bases, bucket_groups=1) as ctx:
            for kind in (0, 1):
                sc = cref.gen_scalars(cid, 7 * n + kind, n, kind=kind)
                assert msm.compress(cid, ctx.msm(sc)) == cref.compress(cid, cref.msm_pippenger(cid, bases, sc, threads=2)), (n, kind)
            canon = np.array([limbs(v) for v in ([0, 1, C.order - 1, 15, 16, (1 << 252) + 5] * n)[:n]], dtype=np.uint64)
            if n >= 8:
                canon[4] = canon[6] = canon[7] = limbs(0x1234567)    # s*P + s*P + s*(-P)
            exp = cref.compress(cid, cref.msm_pippenger(cid, bases, canon, mont=False, threads=2))
            assert msm.compress(cid, ctx.msm(canon, is_mont=False)) == exp, n
            m = max(1, n // 2)
            sc = cref.gen_scalars(cid, 99, m)
            assert msm.compress(cid, ctx.msm(sc)) == cref.compress(cid, cref.msm_pippenger(cid, bases[:m].copy(), sc, threads=2)), ("prefix", n)
            dsc = msm.DeviceBuffer.from_host(sc)
            dout = msm.DeviceBuffer(96)
            ctx.msm(dsc, m, out=dout)                        # device in, device out: nothing but launches
            ctx.sync()
            assert msm.compress(cid, dout.to_host((1, 12))) == cref.compress(cid, cref.msm_pippenger(cid, bases[:m].copy(), sc, threads=2))
            # v*G + b*H: the commitment with a blind, twice with one generator (cached table) and once with another
            for hh, seed in ((h, 1), (h, 2), (h2, 3)):
                v = cref.gen_scalars(cid, 500 + seed, n)
                b = cref.gen_scalars(cid, 600 + seed, 1)
                want = cref.compress(cid, cref.row_msm(cid, bases, v, 1, n, h=hh, blinds=b, threads=2))
                assert msm.compress(cid, ctx.msm_rows(v, 1, n, blinds=b, h=hh)) == want, (n, seed)
            assert (ctx.msm(np.zeros((0, 4), dtype=np.uint64)) == 0).all()
    # many rows with blinds: the blind terms come from H's nibble table too (k_add_blind_tab)
    rows, row_len = 37, 200
    bases = cref.gen_bases_ap(cid, 77, 13, row_len)
    sc = cref.gen_scalars(cid, 31337, rows * row_len, kind=2, small_bound=131)
    bl = cref.gen_scalars(cid, 4, rows)
    bl[3] = 0
    with msm.MsmContext(cid, bases) as ctx:
        assert msm.compress(cid, ctx.msm_rows(sc, rows, row_len, blinds=bl, h=h)) == cref.compress(cid, cref.row_msm(cid, bases, sc, rows, row_len, h=h, blinds=bl, threads=4))


@pytest.mark.parametrize("name,logn,kind,groups", [("pallas", 20, 0, 0), ("pallas", 20, 1, 0), ("vesta", 18, 0, 0),
                                                   ("pallas", 18, 0, 1), ("vesta", 17, 1, 1),
                                                   ("pallas", 20, 0, 1), ("pallas", 20, 1, 1)])   # the last two: bench.py's plan
def test_full_size_dlog_property(name, logn, kind, groups, gpu_lib):
    """BASELINE.json configs[1] size (2^20 Pallas): device-generated bases in arithmetic
    progression, so the result must equal (sum_i s_i*(k0 + i*d)) * G -- a size-independent check
    that needs only O(n) big-int work on the host."""
    from reef_amd import msm
    C = CURVES[name]
    n = 1 << logn
    k0, d = 0xABCDEF, 0x12345
    bases = msm.gen_bases(name, k0, d, n, device=True)
    sc_dev = msm.gen_scalars(name, 0x5EEF, n, kind=kind, mont=True, device=True)
    sc = sc_dev.to_host((n, 4))
    canon = msm.gen_scalars(name, 0x5EEF, n, kind=kind, mont=False)
    with msm.MsmContext(name, bases, n, bucket_groups=groups) as ctx:
        if (logn, groups) == (20, 1):         # what bench.py measures: c = 17, one bucket group, 16 pre-shifted tables, two-level sort
            assert ctx.plan() == {"window_bits": 17, "windows": 16, "bucket_groups": 1, "tables": 16}
        r_dev = ctx.msm(sc_dev, n)            # device-resident scalars
        r_host = ctx.msm(sc)                  # host scalars through the same key
        d_out = msm.DeviceBuffer(96)          # and the result LEFT ON THE DEVICE (a plain key's window combine then runs as a host function on the stream)
        ctx.msm(sc_dev, n, out=d_out)
        ctx.sync()
        r_left = d_out.to_host((12,))
        d_out.free()
    cols = [canon[:, j].astype(object) for j in range(4)]
    idx = np.arange(n, dtype=object)
    acc = 0
    for j in range(4):
        acc += (int(np.sum(cols[j])) * k0 + int(np.sum(cols[j] * idx)) * d) << (64 * j)
    exp = C.compress(C.mul(acc % C.order, C.gen))
    assert msm.compress(name, r_host) == exp
    assert msm.compress(name, r_dev) == exp
    assert msm.compress(name, r_left) == exp
    # Montgomery-form input really is the Montgomery image of the canonical one
    assert C.scalar_from_mont(int.from_bytes(sc[12345].tobytes(), "little")) == int.from_bytes(canon[12345].tobytes(), "little")


@pytest.mark.parametrize("name,n,kind,bound", [("pallas", (1 << 17) + 12345, 2, 1000), ("vesta", 1 << 18, 2, 3), ("pallas", (1 << 19) - 7, 2, 1 << 40),
                                               ("vesta", (1 << 17) + 1, 0, 0), ("pallas", 1 << 18, 1, 0)])
def test_two_level_sort_with_sparse_and_ragged_digits(name, n, kind, bound, gpu_lib):
    """The sort of large inputs (two levels, one-word entries: round 4) on digit sets that stress where k_accum0 finds its bucket
    keys: scalars below a small bound (a handful of giant buckets, every other bucket EMPTY: the key of a run is looked up past
    runs of empty buckets), ragged lengths (the last chunk is short), witness-like and uniform scalars; full-precompute keys as
    the bench uses them and one bucket group per window.  Expected value: the discrete-log closed form."""
    from reef_amd import msm
    C = CURVES[name]
    k0, d = 0x1234567, 0x891
    bases = msm.gen_bases(name, k0, d, n, device=True)
    sc_dev = msm.gen_scalars(name, 0xBEE5, n, kind=kind, small_bound=bound, mont=True, device=True)
    canon = msm.gen_scalars(name, 0xBEE5, n, kind=kind, small_bound=bound, mont=False)
    cols = [canon[:, j].astype(object) for j in range(4)]
    idx = np.arange(n, dtype=object)

    def dlog(upto):
        acc = 0
        for j in range(4):
            acc += (int(np.sum(cols[j][:upto])) * k0 + int(np.sum(cols[j][:upto] * idx[:upto])) * d) << (64 * j)
        return C.compress(C.mul(acc % C.order, C.gen))
    for groups, wbits in ((1, 0), (1, 17), (0, 16)):
        with msm.MsmContext(name, bases, n, bucket_groups=groups, window_bits=wbits) as ctx:
            assert msm.compress(name, ctx.msm(sc_dev, n)) == dlog(n), (groups, wbits)
            m = n - 54321
            assert msm.compress(name, ctx.msm(sc_dev, m)) == dlog(m), (groups, wbits, "prefix")


@pytest.mark.parametrize("n,c", [(61439, 13), (61440, 15), (98304, 15), (180223, 15), (180224, 17)])
def test_shipped_plan_either_side_of_the_window_thresholds_dlog_property(n, c, gpu_lib):
    """choose_window (engine.inc; thresholds moved in round 5 on tools/sweep_window_mid.py's evidence): the plan a pre-shifted key gets
    at the sizes where the window changes -- the longest one-level sort at c = 13 (61 439 x 20 entries), the two-level sort at c = 15
    up to 180 223 points, c = 17 beyond -- uniform, witness-like and tiny scalars, a ragged prefix; expected: the discrete-log closed form."""
    from reef_amd import msm
    C = CURVES["pallas"]
    k0, d = 0x7654321, 0x10F
    bases = msm.gen_bases("pallas", k0, d, n, device=True)
    idx = np.arange(n, dtype=object)
    with msm.MsmContext("pallas", bases, n, bucket_groups=1) as ctx:
        assert ctx.plan()["window_bits"] == c
        for kind, bound in ((0, 0), (1, 0), (2, 5)):
            sc_dev = msm.gen_scalars("pallas", 0xC0DE + kind, n, kind=kind, small_bound=bound, mont=True, device=True)
            canon = msm.gen_scalars("pallas", 0xC0DE + kind, n, kind=kind, small_bound=bound, mont=False)
            cols = [canon[:, j].astype(object) for j in range(4)]
            for upto in (n, n - 4321):
                acc = 0
                for j in range(4):
                    acc += (int(np.sum(cols[j][:upto])) * k0 + int(np.sum(cols[j][:upto] * idx[:upto])) * d) << (64 * j)
                assert msm.compress("pallas", ctx.msm(sc_dev, upto)) == C.compress(C.mul(acc % C.order, C.gen)), (kind, upto)


def test_linearity_and_clone_threads(gpu_lib, cref):
    """MSM(a) + MSM(b) == MSM(a+b); clones of one key used from several threads agree."""
    from reef_amd import msm
    cid = 0
    C = CURVES["pallas"]
    n = 20000
    bases = cref.gen_bases_ap(cid, 123, 11, n)
    a = cref.gen_scalars(cid, 10, n, mont=False)
    b = cref.gen_scalars(cid, 11, n, mont=False)
    s = np.zeros_like(a)
    for i in range(n):
        v = (cref.limbs_to_int(a[i]) + cref.limbs_to_int(b[i])) % C.order
        s[i] = limbs(v)
    with msm.MsmContext(cid, bases, bucket_groups=2) as ctx:
        ra, rb, rs = ctx.msm(a, is_mont=False), ctx.msm(b, is_mont=False), ctx.msm(s, is_mont=False)
        total = msm.sum_points(cid, np.stack([ra, rb]))
        assert msm.compress(cid, total) == msm.compress(cid, rs)
        clones = [ctx.clone() for _ in range(3)]
        results = [None] * 3

        def work(j):
            for _ in range(3):
                results[j] = msm.compress(cid, clones[j].msm(s, is_mont=False))
        ts = [threading.Thread(target=work, args=(j,)) for j in range(3)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert all(r == msm.compress(cid, rs) for r in results)
        for c in clones:
            c.close()


def test_error_paths(gpu_lib, cref):
    from reef_amd import msm
    bases = cref.gen_bases_ap(0, 1, 1, 16)
    with msm.MsmContext(0, bases) as ctx:
        with pytest.raises(ValueError):
            ctx.msm(np.zeros((17, 4), dtype=np.uint64))      # longer than the key: the reference panics
        assert (ctx.msm(np.zeros((0, 4), dtype=np.uint64)) == 0).all()  # empty MSM = identity (0,0,0)
    with pytest.raises(msm.ReefError):
        msm.MsmContext(0, bases, window_bits=31)
    with pytest.raises(ValueError):
        msm.mult_pippenger(0, bases, np.zeros((3, 4), dtype=np.uint64))


# ------------------------------------------------------------------ K2 rows / K3 / K4 ----
def test_rows_golden(golden, gpu_lib, cref):
    from reef_amd import msm
    for case in golden["rows"]:
        C = CURVES[case["curve"]]
        cid = CID[case["curve"]]
        rows, row_len = case["rows"], case["row_len"]
        bases = cref.gen_bases_ap(cid, case["k0"], case["d"], row_len)
        sc = np.array([[s, 0, 0, 0] for s in case["scalars"]], dtype=np.uint64)
        bl = np.array([limbs(int(b, 16)) for b in case["blinds"]], dtype=np.uint64)
        h = np.frombuffer(C.affine_to_bytes(C.mul(case["h_k"], C.gen)), dtype=np.uint64).copy()
        with msm.MsmContext(cid, bases) as ctx:
            for bits in (0, 3, 8):
                out = ctx.msm_rows(sc, rows, row_len, is_mont=False, max_scalar_bits=bits, blinds=bl, h=h)
                comp = msm.compress(cid, out)
                assert [comp[32 * i:32 * i + 32].hex() for i in range(rows)] == case["expect_compressed"], bits


@pytest.mark.parametrize("name,rows,row_len,bound", [("pallas", 64, 512, 7), ("pallas", 32, 2048, 131), ("vesta", 16, 300, 131),
                                                      ("pallas", 8, 1000, 0), ("pallas", 1, 700, 7)])
def test_rows_vs_c_oracle(name, rows, row_len, bound, gpu_lib, cref):
    """Hyrax-shaped commits (commitment.rs:187): DNA-like (<7), ASCII-like (<131) and full-width rows."""
    from reef_amd import msm
    cid = CID[name]
    bases = cref.gen_bases_ap(cid, 77, 13, row_len)
    kind = 2 if bound else 0
    sc = cref.gen_scalars(cid, 31337, rows * row_len, kind=kind, small_bound=bound)
    bl = cref.gen_scalars(cid, 4, rows)
    h = cref.gen_bases_ap(cid, 0xB11D, 1, 1)[0].copy()
    exp = cref.compress(cid, cref.row_msm(cid, bases, sc, rows, row_len, h=h, blinds=bl, threads=4))
    with msm.MsmContext(cid, bases) as ctx:
        out = ctx.msm_rows(sc, rows, row_len, blinds=bl, h=h)
        assert msm.compress(cid, out) == exp
        exp_nb = cref.compress(cid, cref.row_msm(cid, bases, sc, rows, row_len, threads=4))
        assert msm.compress(cid, ctx.msm_rows(sc, rows, row_len)) == exp_nb


@pytest.mark.parametrize("name,rows,row_len,groups,kind", [("pallas", 3, 8192, 0, 0), ("vesta", 2, 9000, 1, 0),
                                                          ("pallas", 5, 8200, 1, 1), ("pallas", 2, 16384, 4, 0),
                                                          ("pallas", 2, 65536, 1, 0), ("vesta", 3, 40000, 1, 1)])   # the last two: two-level sort
def test_long_rows_batched_slice_sort(name, rows, row_len, groups, kind, gpu_lib, cref):
    """Rows of K1 size (>= 8192 wide scalars) go through the sliced sort as one batch of MSMs."""
    from reef_amd import msm
    cid = CID[name]
    bases = cref.gen_bases_ap(cid, 5, 3, row_len)
    sc = cref.gen_scalars(cid, 99, rows * row_len, kind=kind)
    bl = cref.gen_scalars(cid, 5, rows)
    h = cref.gen_bases_ap(cid, 0xB11D, 1, 1)[0].copy()
    exp = cref.compress(cid, cref.row_msm(cid, bases, sc, rows, row_len, h=h, blinds=bl, threads=8))
    with msm.MsmContext(cid, bases, bucket_groups=groups) as ctx:
        assert msm.compress(cid, ctx.msm_rows(sc, rows, row_len, blinds=bl, h=h)) == exp
        assert msm.compress(cid, ctx.msm_rows(sc, rows, row_len, blinds=bl, h=h, max_scalar_bits=255)) == exp


@pytest.mark.parametrize("name,rows,row_len,bound", [("pallas", 700, 300, 7), ("pallas", 1200, 200, 131), ("vesta", 900, 257, 4),
                                                      ("pallas", 600, 1000, 16), ("pallas", 300, 2048, 2), ("vesta", 2500, 100, 256)])
def test_rows_symbol_tables_vs_c_oracle(name, rows, row_len, bound, gpu_lib, cref):
    """Document symbols through the block tables (no bucket method): large enough batches that
    reef_msm_rows takes that path itself, and the same commitments from one-byte symbols
    (reef_msm_rows_symbols); row lengths that are not multiples of the block size; with blinds."""
    from reef_amd import msm
    cid = CID[name]
    bases = cref.gen_bases_ap(cid, 61, 17, row_len)
    sc = cref.gen_scalars(cid, 2718, rows * row_len, kind=2, small_bound=bound)
    canon = cref.gen_scalars(cid, 2718, rows * row_len, kind=2, small_bound=bound, mont=False)
    sym = np.ascontiguousarray(canon[:, 0].astype(np.uint8))
    assert int(canon[:, 0].max()) < bound and not canon[:, 1:].any()
    bl = cref.gen_scalars(cid, 6, rows)
    h = cref.gen_bases_ap(cid, 0xB11D, 1, 1)[0].copy()
    exp = cref.compress(cid, cref.row_msm(cid, bases, sc, rows, row_len, h=h, blinds=bl, threads=8))
    exp_nb = cref.compress(cid, cref.row_msm(cid, bases, sc, rows, row_len, threads=8))
    bits = max(1, (bound - 1).bit_length())
    with msm.MsmContext(cid, bases) as ctx:
        assert msm.compress(cid, ctx.msm_rows(sc, rows, row_len, blinds=bl, h=h)) == exp                 # width measured on the device
        assert msm.compress(cid, ctx.msm_rows(sc, rows, row_len, max_scalar_bits=bits)) == exp_nb        # cached tables
        assert msm.compress(cid, ctx.msm_rows_symbols(sym, rows, row_len, bits, blinds=bl, h=h)) == exp
        if bits < 8:                                                                                   # a wider declared width: other tables
            assert msm.compress(cid, ctx.msm_rows_symbols(sym, rows, row_len, bits + 1)) == exp_nb
        dsym = msm.DeviceBuffer.from_host(sym)
        dout = msm.DeviceBuffer(96 * rows)
        ctx.msm_rows_symbols(dsym, rows, row_len, bits, out=dout)
        ctx.sync()
        assert msm.compress(cid, dout.to_host((rows, 12))) == exp_nb
        assert msm.compress(cid, ctx.msm_rows_symbols(sym[:row_len], 1, row_len, bits)) == exp_nb[:32]   # a single row
        with pytest.raises(msm.ReefError):
            ctx.msm_rows_symbols(sym, rows, row_len, 9)


def test_fold_golden_and_oracle(golden, gpu_lib, cref):
    from reef_amd import msm
    for case in golden["fold"]:
        cid = CID[case["curve"]]
        gens = cref.gen_bases_ap(cid, case["k0"], case["d"], case["n"])
        out = msm.fold(cid, gens, case["n"] // 2, int(case["w1"], 16), int(case["w2"], 16))
        jac = np.zeros((out.shape[0], 12), dtype=np.uint64)
        jac[:, :8] = out
        jac[:, 8:] = cref.field_op("to_mont", cid, cref.int_to_limbs(1))
        comp = cref.compress(cid, jac)
        assert [comp[32 * i:32 * i + 32].hex() for i in range(out.shape[0])] == case["expect_compressed"]
    for cid, name in ((0, "pallas"), (1, "vesta")):
        C = CURVES[name]
        rng = SplitMix64(808)
        gens = cref.gen_bases_ap(cid, 9, 4, 1000)
        gens[3] = gens[503]            # L_i == R_i
        gens[5] = 0                    # identity generator
        w1, w2 = uniform_scalar(rng, C.order), uniform_scalar(rng, C.order)
        assert (msm.fold(cid, gens, 500, w1, w2) == cref.fold(cid, gens, w1, w2)).all()
        assert (msm.fold(cid, gens, 500, w1, w1) == cref.fold(cid, gens, w1, w1)).all()
        # scalars at the edges of the endomorphism split (k = k1 + k2*lambda, 128-bit parts): zero parts, sign
        # changes, r - 1, small values, powers of two; R_i = -L_i and R_i = L_i make sums collapse inside the table
        gens[7] = gens[507]
        gens[508] = np.frombuffer(C.affine_to_bytes(C.neg(C.affine_from_bytes(gens[8].tobytes()))), dtype=np.uint64)
        gens[509] = 0
        r = C.order
        for a, b in ((0, 0), (0, 7), (7, 0), (1, 1), (r - 1, 1), (r - 1, r - 1), (1 << 127, 1 << 128), ((1 << 254) - 3, 5),
                     (uniform_scalar(rng, r), 0), (3, uniform_scalar(rng, r))):
            assert (msm.fold(cid, gens, 500, a, b) == cref.fold(cid, gens, a, b)).all(), (name, hex(a), hex(b))


def test_normalize_matches_oracle(gpu_lib, cref):
    from reef_amd import msm
    for cid in (0, 1):
        n = 1003
        bases = cref.gen_bases_ap(cid, 2, 3, n)
        jac = np.zeros((n, 12), dtype=np.uint64)
        for i in range(n):
            jac[i] = cref.scalar_mul(cid, bases[i], 3 + i)   # Z != 1
        jac[10] = 0                                          # identity (0,0,0)
        aff, comp = msm.normalize(cid, jac, affine=True, compressed=True)
        assert (aff == cref.to_affine(cid, jac)).all()
        assert comp.tobytes() == cref.compress(cid, jac)


@pytest.mark.parametrize("rows,row_len,bound", [(1024, 2048, 131), (512, 4096, 7), (4096, 8192, 7)])
def test_rows_baseline_size_dlog_property(rows, row_len, bound, gpu_lib):
    """HyraxPC::commit (src/backend/commitment.rs:173-187) at BASELINE.json configs[2] size (1 MiB ASCII document =
    1024 rows x 2048 symbols) and configs[3] size (16 MiB DNA = 4096 rows x 8192 symbols < 7): bases in arithmetic
    progression, so row r must be (sum_j Z[r,j]*(k0 + j*d))*G.  Both entry points: field-element scalars
    (reef_msm_rows) and the document's own one-byte symbols (reef_msm_rows_symbols)."""
    from reef_amd import msm
    C = CURVES["pallas"]
    k0, d = 1234567, 89
    bases = msm.gen_bases("pallas", k0, d, row_len, device=True)
    sc = msm.gen_scalars("pallas", 0xD0C, rows * row_len, kind=2, small_bound=bound, device=True)
    canon = msm.gen_scalars("pallas", 0xD0C, rows * row_len, kind=2, small_bound=bound, mont=False)
    assert not canon[:, 1:].any()
    sym = np.ascontiguousarray(canon[:, 0].astype(np.uint8))
    del canon
    out = msm.DeviceBuffer(96 * rows)
    out_sym = msm.DeviceBuffer(96 * rows)
    with msm.MsmContext("pallas", bases, row_len) as ctx:
        ctx.msm_rows(sc, rows, row_len, out=out)
        dsym = msm.DeviceBuffer.from_host(sym)
        ctx.msm_rows_symbols(dsym, rows, row_len, max(1, (bound - 1).bit_length()), out=out_sym)
        ctx.sync()
    comp = msm.compress("pallas", out.to_host((rows, 12)))
    assert msm.compress("pallas", out_sym.to_host((rows, 12))) == comp
    mat = sym.reshape(rows, row_len).astype(np.int64)
    w = k0 + np.arange(row_len, dtype=np.int64) * d             # < 2^21; symbols < 2^8; 2^13 terms: far below 2^63
    for r in list(range(0, rows, 97)) + [rows - 1]:
        acc = int((mat[r] * w).sum()) % C.order
        assert comp[32 * r:32 * r + 32] == C.compress(C.mul(acc, C.gen)), r


@pytest.mark.parametrize("groups,world", [(1, 2), (1, 8), (0, 3), (4, 4)])
def test_window_split_partials_sum_to_the_msm(groups, world, gpu_lib, cref):
    """One MSM split by Pippenger window over `world` devices (north_star): here the ranks
    run one after the other on the same GPU; the partial sums must add up to the whole MSM, for
    pre-shifted and plain keys, single MSMs and rows."""
    from reef_amd import msm
    cid = 0
    n = 3000
    bases = cref.gen_bases_ap(cid, 9, 2, n)
    sc = cref.gen_scalars(cid, 1234, n, kind=0)
    want = cref.compress(cid, cref.msm_pippenger(cid, bases, sc, threads=4))
    rows, row_len = 3, 1000
    bl = cref.gen_scalars(cid, 77, rows)
    h = cref.gen_bases_ap(cid, 0xB11D, 1, 1)[0].copy()
    want_rows = cref.compress(cid, cref.row_msm(cid, bases[:row_len].copy(), sc, rows, row_len, h=h, blinds=bl, threads=4))   # the blind term once
    small = cref.gen_scalars(cid, 3, 600 * 500, kind=2, small_bound=7)                       # large enough for the symbol tables
    want_small = cref.compress(cid, cref.row_msm(cid, bases[:500].copy(), small, 600, 500, h=h, blinds=cref.gen_scalars(cid, 78, 600), threads=8))
    with msm.MsmContext(cid, bases, bucket_groups=groups) as ctx:
        parts, row_parts, small_parts = [], [], []
        for r in range(world):
            ctx.set_window_split(r, world)
            c2 = ctx.clone()                                   # clones inherit the split
            parts.append(c2.msm(sc))
            row_parts.append(ctx.msm_rows(sc, rows, row_len, blinds=bl, h=h))
            small_parts.append(ctx.msm_rows(small, 600, 500, blinds=cref.gen_scalars(cid, 78, 600), h=h))
            c2.close()
        total = msm.sum_points(cid, np.stack(parts))
        assert msm.compress(cid, total) == want
        got_rows = b"".join(msm.compress(cid, msm.sum_points(cid, np.stack([rp[i] for rp in row_parts]))) for i in range(rows))
        assert got_rows == want_rows
        got_small = b"".join(msm.compress(cid, msm.sum_points(cid, np.stack([sp[i] for sp in small_parts]))) for i in range(600))
        assert got_small == want_small
        if world > 1:
            assert msm.compress(cid, parts[0]) != want          # a partial sum, not the MSM
        ctx.set_window_split(0, 1)
        assert msm.compress(cid, ctx.msm(sc)) == want
        with pytest.raises(msm.ReefError):
            ctx.set_window_split(2, 2)


def test_bench_collective_path_on_one_gpu(gpu_lib):
    """bench.py's N > 1 code path (RCCL all_gather of the 96-byte partial sums on the MSM's own HIP
    stream + on-device combine) with a process group of one rank, small size: the JSON line must
    carry a passing parity check."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--exercise-collective", "--logn", "14", "--steps", "4",
                          "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["config"]["check"] == "dlog-ok" and line["n_gpus"] == 1
    assert line["roofline"]["achieved"] > 0 and line["roofline"]["kernel_ms"] > 0




# ---------------------------------------------------------------- byte tables (mid-size resident keys) ----
@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_byte_tables_match_oracle(name, gpu_lib, cref):
    """The sort-free path over byte tables (bucket_groups = 1, byte_tables = 1): every scalar shape, ragged lengths, both
    scalar conventions, identity points, colliding points, and scalars whose bytes sit on the recoding edges."""
    from reef_amd import msm
    cid = CID[name]
    C = CURVES[name]
    rng = SplitMix64(2718)
    n = 3001
    bases = cref.gen_bases_ap(cid, 77, 5, n)
    bases[7] = 0
    bases[100] = bases[101]                                                     # P + P inside one window
    bases[200] = np.frombuffer(C.affine_to_bytes(C.neg(C.affine_from_bytes(bases[201].tobytes()))), dtype=np.uint64)
    with msm.MsmContext(cid, bases, bucket_groups=1, byte_tables=1) as ctx:
        assert ctx.has_byte_tables()
        for kind, bound in ((0, 0), (1, 0), (2, 131)):
            sc = cref.gen_scalars(cid, 5 + kind, n, kind=kind, small_bound=bound)
            sc[100] = sc[101]
            sc[200] = sc[201]
            for m in (n, 1, 2, 1025, 2999):
                want = cref.compress(cid, cref.msm_pippenger(cid, bases[:m].copy(), sc[:m].copy()))
                assert msm.compress(cid, ctx.msm(sc[:m].copy())) == want, (kind, m)
            canon = cref.gen_scalars(cid, 5 + kind, n, kind=kind, small_bound=bound, mont=False)
            canon[100] = canon[101]
            canon[200] = canon[201]
            assert msm.compress(cid, ctx.msm(canon, is_mont=False)) == cref.compress(cid, cref.msm_pippenger(cid, bases, sc))
        # CE::commit(v, blind) on the same key: v*G from the byte tables + blind*H
        sc = cref.gen_scalars(cid, 21, n)
        bl = cref.gen_scalars(cid, 22, 1)
        h = cref.gen_bases_ap(cid, 0xB11D, 1, 1)[0].copy()
        want = cref.compress(cid, cref.row_msm(cid, bases, sc, 1, n, h=h, blinds=bl, threads=4))
        assert msm.compress(cid, ctx.msm_rows(sc, 1, n, blinds=bl, h=h)) == want
        # bytes on the edges of the signed recoding: 0x80 (kept), 0x81 (negative with a carry), 0xff runs (carry chains), r - 1
        r = C.order
        edge = [0, 1, 0x80, 0x81, 0xFF, 0x100, 0x7F80, 0x8080, 0x80FF, (1 << 254) - 1, (1 << 254), r - 1, r - 2,
                int("80" * 31, 16), int("81" * 31, 16), int("7f" + "ff" * 30, 16), int("ff" * 31, 16) % r]
        vals = [edge[j % len(edge)] if j < 4 * len(edge) else uniform_scalar(rng, r) for j in range(n)]
        sc = np.array([[(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)] for v in vals], dtype=np.uint64)
        acc = None
        pts = [C.affine_from_bytes(bases[i].tobytes()) for i in range(4 * len(edge))]
        for v, p in zip(vals, pts):
            acc = C.add(acc, C.mul(v, p))
        got = ctx.msm(sc[:4 * len(edge)].copy(), is_mont=False)
        assert msm.compress(cid, got) == C.compress(acc)


@pytest.mark.parametrize("name,logn,kind", [("pallas", 15, 0), ("pallas", 16, 0), ("vesta", 14, 1), ("pallas", 16, 1)])
def test_byte_tables_at_prover_sizes_dlog_property(name, logn, kind, gpu_lib):
    """Reef's own key sizes (2^14 .. 2^16) through the byte tables: bases in arithmetic progression, so the result is
    (sum_i s_i*(k0 + i*d)) * G; the bucket pipeline on the same key (byte_tables = 2) must give the same point."""
    from reef_amd import msm
    C = CURVES[name]
    n = 1 << logn
    k0, d = 0xFEDCBA, 0x7531
    bases = msm.gen_bases(name, k0, d, n, device=True)
    sc_dev = msm.gen_scalars(name, 0xACE, n, kind=kind, mont=True, device=True)
    canon = msm.gen_scalars(name, 0xACE, n, kind=kind, mont=False)
    with msm.MsmContext(name, bases, n, bucket_groups=1, byte_tables=1) as ctx, msm.MsmContext(name, bases, n, bucket_groups=1, byte_tables=2) as ref:
        assert ctx.has_byte_tables() and not ref.has_byte_tables()
        got = ctx.msm(sc_dev, n)
        assert msm.compress(name, got) == msm.compress(name, ref.msm(sc_dev, n))
        m = n - 12345                                                          # a ragged prefix of the key
        got_m = ctx.msm(sc_dev, m)
    cols = [canon[:, j].astype(object) for j in range(4)]
    idx = np.arange(n, dtype=object)

    def dlog(upto):
        acc = 0
        for j in range(4):
            acc += (int(np.sum(cols[j][:upto])) * k0 + int(np.sum(cols[j][:upto] * idx[:upto])) * d) << (64 * j)
        return C.compress(C.mul(acc % C.order, C.gen))
    assert msm.compress(name, got) == dlog(n)
    assert msm.compress(name, got_m) == dlog(m)


def test_byte_tables_built_in_the_background_on_request(gpu_lib, cref):
    """byte_tables = 3: the tables of an eligible key are built on a stream of their own from reef_msm_ctx_create on; the bucket
    pipeline serves the key meanwhile, the switch changes no result, clones share the tables, re-keying drops them (a build in
    flight is told to stop: the levels still queued return at once).  The DEFAULT (byte_tables = 0) builds none: the tables cost
    256 KiB per point and their build slows the calls it runs beside, so a key gets them only when its owner asks."""
    import os
    import time
    from reef_amd import msm
    if os.environ.get("REEF_MSM_WIDE") == "0":
        pytest.skip("byte tables switched off by the environment")
    cid, n = 0, 16384
    bases = cref.gen_bases_ap(cid, 9, 2, n)
    sc = cref.gen_scalars(cid, 4, n)
    want = cref.compress(cid, cref.msm_pippenger(cid, bases, sc, threads=8))
    if os.environ.get("REEF_MSM_WIDE") != "1":
        with msm.MsmContext(cid, bases, bucket_groups=1) as dflt:            # the default policy: no tables, whatever the key has served
            for _ in range(3):
                assert msm.compress(cid, dflt.msm(sc)) == want
            time.sleep(0.2)
            assert not dflt.has_byte_tables()
    with msm.MsmContext(cid, bases, bucket_groups=1, byte_tables=3) as ctx:
        clone = ctx.clone()
        seen = set()
        deadline = time.time() + 20
        while time.time() < deadline:                                         # calls before, during and after the build
            seen.add(ctx.has_byte_tables())
            assert msm.compress(cid, (ctx if len(seen) % 2 else clone).msm(sc)) == want
            if True in seen:
                break
        assert True in seen, "the tables never became ready"
        assert clone.has_byte_tables()
        assert msm.compress(cid, clone.msm(sc)) == want and msm.compress(cid, ctx.msm(sc[:5000].copy())) == cref.compress(
            cid, cref.msm_pippenger(cid, bases[:5000].copy(), sc[:5000].copy()))
        clone.close()
        bases2 = cref.gen_bases_ap(cid, 77, 3, n)                              # re-keying: the old tables must not serve the new key
        ctx.set_bases(bases2)
        want2 = cref.compress(cid, cref.msm_pippenger(cid, bases2, sc, threads=8))
        assert msm.compress(cid, ctx.msm(sc)) == want2
        t0 = time.time()
        for k in range(4):                                                     # re-keying in a loop: every call stops the build the previous one started
            ctx.set_bases(bases if k % 2 == 0 else bases2)
            assert msm.compress(cid, ctx.msm(sc)) == (want if k % 2 == 0 else want2)
        assert time.time() - t0 < 5.0
    with msm.MsmContext(cid, bases, bucket_groups=1, byte_tables=2) as none:
        assert msm.compress(cid, none.msm(sc)) == want and not none.has_byte_tables()
    with msm.MsmContext(cid, bases, bucket_groups=1, byte_tables=3) as gone:               # destroyed while the build is in flight
        pass
    with msm.MsmContext(cid, bases[:1000].copy(), bucket_groups=1, byte_tables=1) as small:   # <= 1024 points: the nibble tables serve it
        assert not small.has_byte_tables()
    with msm.MsmContext(cid, bases, bucket_groups=0, byte_tables=1) as plain:               # not pre-shifted: no tables
        assert not plain.has_byte_tables()


def test_byte_tables_switch_under_concurrent_callers(gpu_lib, cref):
    """Four caller threads on clones of one key issue MSMs from the moment the key exists, through the background build and
    the switch to the byte tables: every result is the same point."""
    import time
    from reef_amd import msm
    cid, n = 1, 30000
    bases = cref.gen_bases_ap(cid, 314, 15, n)
    sc = cref.gen_scalars(cid, 27, n)
    want = cref.compress(cid, cref.msm_pippenger(cid, bases, sc, threads=8))
    with msm.MsmContext(cid, bases, bucket_groups=1, byte_tables=3) as ctx:
        clones = [ctx.clone() for _ in range(4)]
        bad, calls = [], [0] * 4
        stop = time.time() + 1.5

        def work(j):
            while time.time() < stop or not clones[j].has_byte_tables():
                if msm.compress(cid, clones[j].msm(sc)) != want:
                    bad.append(j)
                calls[j] += 1
                if time.time() > stop + 20:
                    bad.append(("never ready", j))
                    break
            for _ in range(3):                                                 # and a few calls after the switch
                if msm.compress(cid, clones[j].msm(sc)) != want:
                    bad.append(("after", j))
        ts = [threading.Thread(target=work, args=(j,)) for j in range(4)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert not bad, bad
        assert min(calls) > 3 and ctx.has_byte_tables()
        for c in clones:
            c.close()


def test_captured_graph_path_gives_the_same_points(gpu_lib):
    """REEF_MSM_GRAPH=1 (opt-in, read once per process): the bucket pipeline of a repeated call shape is captured as a hipGraph
    the second time it is seen and replayed afterwards; results are those of the plain launches, across re-keying, other
    lengths on the same ctx (workspace growth drops the captured graphs) and host / device scalars."""
    import os
    import subprocess
    import sys
    code = r"""
import sys; sys.path.insert(0, '.')
import numpy as np
from oracle import pasta_ref as R
from reef_amd import msm
for cid in (0, 1):
    n = 6000
    bases = R.gen_bases_ap(cid, 31, 5, n); sc = R.gen_scalars(cid, 8, n); sc2 = R.gen_scalars(cid, 9, n, kind=1)
    want = R.compress(cid, R.msm_pippenger(cid, bases, sc, threads=4)); want2 = R.compress(cid, R.msm_pippenger(cid, bases, sc2, threads=4))
    with msm.MsmContext(cid, bases, bucket_groups=1, byte_tables=2) as ctx:
        for k in range(5):                                   # plain, captured, replayed ...
            assert msm.compress(cid, ctx.msm(sc)) == want, k
            assert msm.compress(cid, ctx.msm(sc2)) == want2, k       # same staging buffer, other contents
        m = 2500
        wantm = R.compress(cid, R.msm_pippenger(cid, bases[:m].copy(), sc[:m].copy(), threads=4))
        for k in range(4):
            assert msm.compress(cid, ctx.msm(sc[:m].copy())) == wantm
        big = R.gen_bases_ap(cid, 77, 3, 20000); scb = R.gen_scalars(cid, 3, 20000)
        ctx.set_bases(big)                                    # re-keyed: captured launches of the old key are gone
        wantb = R.compress(cid, R.msm_pippenger(cid, big, scb, threads=8))
        d = msm.DeviceBuffer.from_host(scb); o = msm.DeviceBuffer(96)
        for k in range(4):
            assert msm.compress(cid, ctx.msm(scb)) == wantb
            ctx.msm(d, 20000, out=o); ctx.sync()
            assert msm.compress(cid, o.to_host(12)) == wantb
print('graph-ok')
"""
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, REEF_MSM_GRAPH="1"), capture_output=True, text=True, timeout=600,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert "graph-ok" in out.stdout, out.stderr[-3000:]


@pytest.mark.parametrize("name", ["pallas", "vesta"])
def test_plain_key_results_that_stay_on_the_device(name, gpu_lib, cref):
    """Round 6: a plain key (no pre-shifted tables: pasta-msm's contract, and what the IPA's folded bases get) with the RESULT LEFT ON THE DEVICE.  The window
    combine -- ~255 dependent doublings -- is finished by a host function enqueued on the context's stream (hipLaunchHostFunc) and copied back in stream
    order, so the call still only enqueues work.  Several calls back to back on one context before anything is waited for (each host function must see
    ITS group sums), device and host scalars, a prefix of the key, window splits, and a window so small that the group sums do not fit the landing zone
    (the on-device Horner kernel takes over)."""
    from reef_amd import msm
    cid = msm.curve_id(name)
    for n, wbits in ((1, 0), (777, 0), (5000, 0), (40000, 0), (3000, 3), (3000, 2)):
        bases = cref.gen_bases_ap(cid, 2024 + n, 3, n)
        bases[n // 2] = 0
        scs = [cref.gen_scalars(cid, 50 + j, n, kind=j % 2) for j in range(4)]
        want = [cref.compress(cid, cref.msm_pippenger(cid, bases, s, threads=8)) for s in scs]
        with msm.MsmContext(cid, bases, bucket_groups=0, window_bits=wbits) as ctx:
            outs = [msm.DeviceBuffer(96) for _ in scs]
            dsc = [msm.DeviceBuffer.from_host(s) for s in scs[:2]]
            for rep in range(2):
                for j in range(4):                                   # enqueued one after the other, nothing waited for in between
                    ctx.msm(dsc[j] if j < 2 else scs[j], n, out=outs[j])
                ctx.sync()
                for j in range(4):
                    assert msm.compress(cid, outs[j].to_host((12,))) == want[j], (n, wbits, rep, j)
            if n > 10:
                m = n - 7
                ctx.msm(scs[1][:m].copy(), m, out=outs[0])
                ctx.sync()
                assert msm.compress(cid, outs[0].to_host((12,))) == cref.compress(cid, cref.msm_pippenger(cid, bases[:m].copy(), scs[1][:m].copy(), threads=8))
                parts = []
                for r in range(3):                                    # window split: the partial sums stay on the device too
                    ctx.set_window_split(r, 3)
                    ctx.msm(dsc[0], n, out=outs[r])
                ctx.sync()
                parts = [outs[r].to_host((12,)) for r in range(3)]
                assert msm.compress(cid, msm.sum_points(cid, np.stack(parts))) == want[0], (n, wbits)
                ctx.set_window_split(0, 1)
            for b in outs + dsc:
                b.free()
