#!/usr/bin/env python3
"""Randomised soak of the C ABI against the oracle's C restatement: sizes, curves, scalar shapes, key kinds,
rows and symbol commits, window splits.  Usage: python tools/soak.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import keygen_oracle as KO, merkle_oracle as MO, mle_oracle, pasta_ref as R, sumcheck_oracle as S  # noqa: E402
from reef_amd import keygen, merkle, mle, msm  # noqa: E402
from reef_amd.sumcheck import SumCheck  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
done = 0
while time.time() < t_end:
    cid = int(rng.integers(0, 2))
    shape = rng.integers(0, 13)
    if shape == 0:      # single MSM, any size
        n = int(2 ** rng.uniform(0, 21.2 if os.environ.get("SOAK_BIG") else 18.5))
        kind = int(rng.integers(0, 3))
        bases = R.gen_bases_ap(cid, int(rng.integers(1, 1 << 30)), int(rng.integers(1, 1000)), n)
        sc = R.gen_scalars(cid, int(rng.integers(1, 1 << 30)), n, kind=kind, small_bound=int(rng.integers(2, 70000)))
        if n > 8 and rng.random() < 0.3:
            bases[rng.integers(0, n, size=3)] = 0
            sc[rng.integers(0, n, size=3)] = 0
        want = R.compress(cid, R.msm_pippenger(cid, bases, sc, threads=16))
        groups = int(rng.choice([0, 1, 1, 3]))
        with msm.MsmContext(cid, bases, bucket_groups=groups, window_bits=int(rng.choice([0, 0, 7, 13, 15, 16])),
                            byte_tables=int(rng.choice([0, 1, 1, 2, 3])) if n <= 65536 else 0) as ctx:
            m = int(rng.integers(1, n + 1)) if rng.random() < 0.3 else n
            got = msm.compress(cid, ctx.msm(sc[:m].copy()))
            if m != n:
                want = R.compress(cid, R.msm_pippenger(cid, bases[:m].copy(), sc[:m].copy(), threads=16))
            assert got == want, ("msm", cid, n, m, kind, groups)
            if rng.random() < 0.3:
                world = int(rng.integers(2, 9))
                parts = []
                for r in range(world):
                    ctx.set_window_split(r, world)
                    parts.append(ctx.msm(sc[:m].copy()))
                ctx.set_window_split(0, 1)
                assert msm.compress(cid, msm.sum_points(cid, np.stack(parts))) == want, ("split", cid, n, m, world)
    elif shape in (11, 12):   # device groups (one process, several members: ordinals repeat beyond the box's devices): MSMs and rows
        n = int(2 ** rng.uniform(0, 17))
        members = int(rng.integers(1, 9))
        devs = [int(i) % msm.device_count() for i in range(members)]
        bases = R.gen_bases_ap(cid, int(rng.integers(1, 1 << 30)), int(rng.integers(1, 1000)), n)
        if n > 8 and rng.random() < 0.3:
            bases[rng.integers(0, n, size=2)] = 0
        sc = R.gen_scalars(cid, int(rng.integers(1, 1 << 30)), n, kind=int(rng.integers(0, 2)))
        split = int(rng.integers(0, 2))
        with msm.MsmGroup(cid, bases, devs, split=split, exchange=int(rng.choice([1, 2, 3])), bucket_groups=int(rng.choice([0, 1, 1]))) as g:
            m = int(rng.integers(0, n + 1)) if rng.random() < 0.3 else n
            want = R.compress(cid, R.msm_pippenger(cid, bases[:m].copy(), sc[:m].copy(), threads=16)) if m else bytes(32)
            buf = sc[:m].copy() if m else np.zeros((0, 4), np.uint64)
            assert msm.compress(cid, g.msm(buf, m)) == want, ("group", cid, n, m, members, split)
            if split == 0 and rng.random() < 0.5:
                rows = int(rng.integers(1, 40))
                row_len = int(rng.integers(1, min(n, 600) + 1))
                bound = int(rng.choice([7, 131, 0]))
                rsc = R.gen_scalars(cid, int(rng.integers(1, 1 << 30)), rows * row_len, kind=2 if bound else 0, small_bound=bound)
                bl = R.gen_scalars(cid, int(rng.integers(1, 1 << 30)), rows)
                h = R.gen_bases_ap(cid, 0xB11D, 1, 1)[0].copy()
                wr = R.compress(cid, R.row_msm(cid, bases[:row_len].copy(), rsc, rows, row_len, h=h, blinds=bl, threads=16))
                assert msm.compress(cid, g.msm_rows(rsc, rows, row_len, blinds=bl, h=h)) == wr, ("group-rows", cid, rows, row_len, members)
    elif shape == 7:    # key derivation (stand-in parameter sets)
        name = "pallas" if cid == 0 else "vesta"
        k = KO.standin_params(name, int(rng.integers(0, 3)), bool(rng.integers(0, 2)))
        label, n = rng.bytes(int(rng.integers(0, 40))), int(rng.integers(1, 48))
        raw = keygen.derive_generators(name, label, n, k.a, k.b, k.z, k.iso, k.dst, k.little_endian)
        assert keygen.points_to_ints(name, raw) == KO.from_label(label, n, k), ("keygen", name, n)
    elif shape == 8:    # Poseidon Merkle tree (stand-in constants, any number of partial rounds)
        p = MO.standin_params(MO.Q if cid == 0 else MO.P, 5, int(rng.choice([2, 4, 8])), int(rng.integers(0, 60)))
        doc = [int(v) for v in rng.integers(0, 1 << 32, size=int(rng.integers(1, 120)), dtype=np.uint64)]
        want = MO.commit(doc, p)
        assert merkle.commit("pallas" if cid == 0 else "vesta", doc, p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node) == want, ("merkle", cid, p.rf, p.rp, len(doc))
        if rng.integers(0, 2):    # the same tree in blocks over a group of members (reef_merkle_commit_devices)
            from reef_amd.sumcheck import array_to_ints
            members = int(rng.integers(1, 9))
            root, tree = merkle.commit_arrays("pallas" if cid == 0 else "vesta", np.asarray(doc, dtype=np.uint32), p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node,
                                              devices=[0] * members)
            assert (root, [array_to_ints(lv) for lv in tree]) == want, ("merkle-blocks", cid, members, len(doc))
    elif shape == 9:    # commitments over folded generators, folds recorded not performed
        logn = int(rng.integers(1, 13))
        n = 1 << logn
        gens0 = R.gen_bases_ap(cid, int(rng.integers(1, 1 << 30)), 7, n)
        order = S.Q if cid == 0 else mle_oracle.P
        k = int(rng.integers(0, logn + 1))
        w1s = [int.from_bytes(rng.bytes(32), "little") % order for _ in range(k)]
        w2s = [int.from_bytes(rng.bytes(32), "little") % order for _ in range(k)]
        gens = gens0
        for a_, b_ in zip(w1s, w2s):
            gens = R.fold(cid, np.ascontiguousarray(gens), a_, b_)
        n_k = n >> k
        off = int(rng.integers(0, n_k))
        ln = int(rng.integers(1, n_k - off + 1))
        v = R.gen_scalars(cid, int(rng.integers(1, 1 << 30)), ln)
        with msm.MsmContext(cid, gens0, bucket_groups=int(rng.choice([0, 1])), byte_tables=int(rng.choice([0, 1, 2, 3]))) as ctx:
            got = msm.compress(cid, ctx.msm_folded(v, w1s, w2s, off))
        assert got == R.compress(cid, R.msm_pippenger(cid, np.ascontiguousarray(gens[off:off + ln]), v)), ("folded", cid, n, k, off, ln)
    elif shape == 10:   # collisions everywhere: few distinct points (and their negatives), few distinct scalars
        n = int(2 ** rng.uniform(1, 14))
        distinct = R.gen_bases_ap(cid, int(rng.integers(1, 1 << 30)), 1, int(rng.integers(1, 4)))
        C = __import__("oracle.pasta_oracle", fromlist=["CURVES"]).CURVES["pallas" if cid == 0 else "vesta"]
        pool = [distinct[i].copy() for i in range(distinct.shape[0])]
        pool += [np.frombuffer(C.affine_to_bytes(C.neg(C.affine_from_bytes(x.tobytes()))), dtype=np.uint64).copy() for x in pool]
        bases = np.stack([pool[int(j)] for j in rng.integers(0, len(pool), size=n)])
        few = R.gen_scalars(cid, int(rng.integers(1, 1 << 30)), 3)
        sc = np.stack([few[int(j)] for j in rng.integers(0, 3, size=n)])
        want = R.compress(cid, R.msm_pippenger(cid, bases, sc, threads=16))
        with msm.MsmContext(cid, bases, bucket_groups=int(rng.choice([0, 1])), window_bits=int(rng.choice([0, 5, 13])), byte_tables=int(rng.choice([1, 2]))) as ctx:
            assert msm.compress(cid, ctx.msm(sc)) == want, ("collide", cid, n)
    elif shape == 1:    # stateless drop-in symbol
        n = int(2 ** rng.uniform(0, 16))
        bases = R.gen_bases_ap(cid, int(rng.integers(1, 1 << 30)), 3, n)
        sc = R.gen_scalars(cid, int(rng.integers(1, 1 << 30)), n, kind=int(rng.integers(0, 2)))
        assert msm.compress(cid, msm.mult_pippenger(cid, bases, sc)) == R.compress(cid, R.msm_pippenger(cid, bases, sc, threads=16)), ("pip", cid, n)
    elif shape == 6:    # generator fold (endomorphism split of both scalars)
        half = int(2 ** rng.uniform(0, 11))
        gens = R.gen_bases_ap(cid, int(rng.integers(1, 1 << 30)), 7, 2 * half)
        if half > 4:
            gens[rng.integers(0, 2 * half, size=2)] = 0
            gens[1] = gens[half + 1]
        order = S.Q if cid == 0 else mle_oracle.P
        pick = lambda: int(rng.choice([0, 1, 2, order - 1, 1 << 127, 1 << 128])) if rng.random() < 0.3 else int.from_bytes(rng.bytes(32), "little") % order  # noqa: E731
        w1, w2 = pick(), pick()
        assert (msm.fold(cid, gens, half, w1, w2) == R.fold(cid, gens, w1, w2)).all(), ("fold", cid, half, hex(w1), hex(w2))
    elif shape == 4:    # document polynomial: bound rows / evaluation
        mod = S.Q if cid == 0 else mle_oracle.P
        m = int(rng.integers(0, 13))
        left = int(rng.integers(0, m + 1))
        n = int(rng.integers(0, (1 << m) + 1))
        point = [int.from_bytes(rng.bytes(32), "little") % mod for _ in range(m)]
        if rng.random() < 0.5:
            z = rng.integers(0, 256, size=n, dtype=np.uint8)
            zi = [int(v) for v in z]
        else:
            zi = [int.from_bytes(rng.bytes(32), "little") % mod for _ in range(n)]
            z = zi
        assert mle.bound_rows("pallas" if cid == 0 else "vesta", z, point, left) == mle_oracle.bound_rows(zi, point, left, mod), ("mle", cid, m, left, n)
    elif shape == 5:    # a whole sum-check (pallas scalar field); half the time with the rank-one EQ rounds and the row structure of
        ell = int(rng.integers(1, 11))                          # the pristine table forced on at this size
        if rng.random() < 0.5:
            os.environ["REEF_SC_RANK1_MIN_POW"] = "1"
        else:
            os.environ.pop("REEF_SC_RANK1_MIN_POW", None)
        cols = 1 << (ell // 2)
        table = []
        for _row in range((1 << ell) // cols):
            pick = int(rng.integers(0, 4))
            if pick == 0:
                table += [int.from_bytes(rng.bytes(32), "little") % S.Q for _ in range(cols)]
            elif pick == 1:
                table += [int(rng.integers(0, 1 << 30)) for _ in range(cols)]
            elif pick == 2:
                table += [int.from_bytes(rng.bytes(32), "little") % S.Q] * cols
            else:
                table += [int(rng.integers(0, 7)) for _ in range(cols)]
        table = table[:int(rng.integers(1 << (ell - 1), (1 << ell) + 1))]
        nq = int(rng.integers(1, 6))
        qs = [int(rng.integers(0, 1 << ell)) for _ in range(nq)]
        rs = [int.from_bytes(rng.bytes(32), "little") % S.Q for _ in range(nq + 1)]
        last_q = [int.from_bytes(rng.bytes(32), "little") % S.Q for _ in range(ell)]
        t = table + [0] * ((1 << ell) - len(table))
        e = S.gen_eq_table(rs, qs, last_q)
        with SumCheck("pallas", ell) as sc:
            sc.set_table(0, table)
            sc.gen_eq_table(rs, qs, last_q)
            fused = rng.random() < 0.5
            g = sc.round_coeffs(1)
            for i in range(1, ell + 1):
                assert g == S.linear_mle_coeffs(t, e, ell, i), ("sc", ell, i, fused, os.environ.get("REEF_SC_RANK1_MIN_POW"))
                r = int.from_bytes(rng.bytes(32), "little") % S.Q
                S.linear_mle_fold(t, e, ell, i, r)
                if fused and i < ell:
                    g = sc.fold_and_next_coeffs(i, r)
                else:
                    sc.fold(i, r)
                    if i < ell:
                        g = sc.round_coeffs(i + 1)
            assert sc.read(0, 1)[0] == t[0] and sc.read(1, 1)[0] == e[0]
        os.environ.pop("REEF_SC_RANK1_MIN_POW", None)
    else:               # rows (field elements and symbols)
        big = 2 if os.environ.get("SOAK_BIG") else 0
        rows, row_len = int(2 ** rng.uniform(0, 11 + big)), int(2 ** rng.uniform(0, 12 + big / 2))
        bound = int(rng.choice([2, 4, 7, 16, 131, 256, 1 << 16, 0]))
        bases = R.gen_bases_ap(cid, int(rng.integers(1, 1 << 30)), 5, row_len)
        seed = int(rng.integers(1, 1 << 30))
        sc = R.gen_scalars(cid, seed, rows * row_len, kind=2 if bound else 0, small_bound=bound)
        bl = R.gen_scalars(cid, seed + 1, rows)
        h = R.gen_bases_ap(cid, 0xB11D, 1, 1)[0].copy()
        want = R.compress(cid, R.row_msm(cid, bases, sc, rows, row_len, h=h, blinds=bl, threads=16))
        with msm.MsmContext(cid, bases, bucket_groups=int(rng.choice([0, 1]))) as ctx:
            assert msm.compress(cid, ctx.msm_rows(sc, rows, row_len, blinds=bl, h=h)) == want, ("rows", cid, rows, row_len, bound)
            if 0 < bound <= 256:
                sym = np.ascontiguousarray(R.gen_scalars(cid, seed, rows * row_len, kind=2, small_bound=bound, mont=False)[:, 0].astype(np.uint8))
                bits = max(1, (bound - 1).bit_length())
                assert msm.compress(cid, ctx.msm_rows_symbols(sym, rows, row_len, bits, blinds=bl, h=h)) == want, ("sym", cid, rows, row_len, bound)
            if rng.random() < 0.5:      # every buffer on the device, another h (its table is checked and rebuilt on the device)
                h2 = R.gen_bases_ap(cid, int(rng.integers(1, 1 << 30)), 1, 1)[0].copy() if rng.random() < 0.8 else np.zeros_like(h)
                want2 = R.compress(cid, R.row_msm(cid, bases, sc, rows, row_len, h=h2, blinds=bl, threads=16))
                d_sc, d_bl, d_h, d_out = (msm.DeviceBuffer.from_host(sc), msm.DeviceBuffer.from_host(bl), msm.DeviceBuffer.from_host(h2),
                                          msm.DeviceBuffer(96 * rows))
                for _ in range(2):
                    ctx.msm_rows(d_sc, rows, row_len, blinds=d_bl, h=d_h, out=d_out)
                    ctx.sync()
                    assert msm.compress(cid, d_out.to_host((rows, 12))) == want2, ("rows-device", cid, rows, row_len, bound)
                assert msm.compress(cid, ctx.msm_rows(sc, rows, row_len, blinds=bl, h=h)) == want, ("rows-host-again", cid, rows, row_len)
    done += 1
from reef_amd import _ffi  # noqa: E402
print(f"soak ok: {done} random cases in {budget:.0f} s on library sources {_ffi.library_sources_sha16()} ({_ffi.load().reef_version().decode()})")
