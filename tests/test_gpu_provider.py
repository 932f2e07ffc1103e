"""The provider mirror (reef_amd/provider.py) exercised the way Reef's call sites use nova-snark:
CE::commit with a blind (commitment.rs:350,361), the hybrid equality relation
(commitment.rs:407-444, test eq_proof :633-680), HyraxPC::commit (commitment.rs:187) and one IPA
round's generator fold + the two cross MSMs (framework.rs:695)."""
import numpy as np
import pytest

from oracle.pasta_oracle import CURVES, SplitMix64, uniform_scalar

pytestmark = pytest.mark.gpu


def limbs(v):
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def test_commit_with_blind_is_homomorphic(gpu_lib, cref):
    """eq_proof shape: C1 = v*G + r1*H, C2 = v*G + r2*H  =>  C1 - C2 = (r1 - r2)*H."""
    from reef_amd import provider as P
    C = CURVES["pallas"]
    G = cref.gen_bases_ap(0, 5, 1, 1)
    H = cref.gen_bases_ap(0, 0xB11D, 1, 1)[0]
    gens = P.CommitmentGens("pallas", G, H)
    rng = SplitMix64(1)
    v, r1, r2 = (uniform_scalar(rng, C.order) for _ in range(3))
    c1 = gens.commit(P.scalars_to_array([v], "pallas"), P.scalars_to_array([r1], "pallas")[0], is_mont=False)
    c2 = gens.commit(P.scalars_to_array([v], "pallas"), P.scalars_to_array([r2], "pallas")[0], is_mont=False)
    g, h = C.affine_from_bytes(G[0].tobytes()), C.affine_from_bytes(H.tobytes())
    assert c1.compress() == C.compress(C.add(C.mul(v, g), C.mul(r1, h)))
    assert c2.compress() == C.compress(C.add(C.mul(v, g), C.mul(r2, h)))
    hgens = P.CommitmentGens("pallas", H.reshape(1, 8))
    diff = hgens.commit(P.scalars_to_array([(r1 - r2) % C.order], "pallas"), is_mont=False)
    neg_c2 = P.Commitment("pallas", c2.jac.copy())
    # -C2: negate y of the affine form and re-wrap
    aff = neg_c2.to_affine()
    pt = C.neg(C.affine_from_bytes(aff.tobytes()))
    j = np.zeros(12, dtype=np.uint64)
    j[:8] = np.frombuffer(C.affine_to_bytes(pt), dtype=np.uint64)
    j[8:] = cref.field_op("to_mont", 0, cref.int_to_limbs(1))
    assert (c1 + P.Commitment("pallas", j)) == diff
    gens.close(); hgens.close()


def test_hyrax_commit_matches_oracle(gpu_lib, cref):
    from reef_amd import provider as P
    for name, cid, bound in (("pallas", 0, 7), ("pallas", 0, 131), ("vesta", 1, 131)):
        num_vars = 13                               # 8192 symbols -> 64 rows x 128 columns
        left, right = P.HyraxPC.compute_factored_lens(num_vars)
        assert (left, right) == (6, 7)
        rows, row_len = 1 << left, 1 << right
        gens_v = P.CommitmentGens(name, cref.gen_bases_ap(cid, 9, 2, row_len), cref.gen_bases_ap(cid, 0xB11D, 1, 1)[0])
        poly = cref.gen_scalars(cid, 7, rows * row_len, kind=2, small_bound=bound)
        blinds = cref.gen_scalars(cid, 8, rows)
        out, comp = P.HyraxPC(gens_v).commit(poly, blinds)
        exp = cref.row_msm(cid, gens_v.bases, poly, rows, row_len, h=gens_v.h, blinds=blinds, threads=4)
        assert comp.tobytes() == cref.compress(cid, exp)
        gens_v.close()


def test_ipa_round_shape(gpu_lib, cref):
    """One round of the inner-product argument as nova's ipa_pc runs it: L = <a_lo, G_hi>,
    R = <a_hi, G_lo> (two MSMs of n/2 points with changing bases), then G' = fold(G, r^-1, r)."""
    from reef_amd import provider as P
    cid, name = 0, "pallas"
    C = CURVES[name]
    n = 512
    gens = P.CommitmentGens(name, cref.gen_bases_ap(cid, 21, 4, n), precompute=False)
    a = cref.gen_scalars(cid, 3, n, mont=False)
    g_lo, g_hi = gens.split_at(n // 2)
    L = g_hi.commit(a[: n // 2].copy(), is_mont=False)
    Rr = g_lo.commit(a[n // 2:].copy(), is_mont=False)
    assert L.compress() == cref.compress(cid, cref.msm_pippenger(cid, g_hi.bases, a[: n // 2].copy(), mont=False))
    assert Rr.compress() == cref.compress(cid, cref.msm_pippenger(cid, g_lo.bases, a[n // 2:].copy(), mont=False))
    r = 0x1234567890ABCDEF1234567890ABCDEF % C.order
    rinv = pow(r, -1, C.order)
    folded = gens.fold(rinv, r)
    assert (folded.bases == cref.fold(cid, gens.bases, rinv, r)).all()
    assert len(g_lo.combine(g_hi)) == n
    # folded commitment relation: <a', G'> with a' = r*a_lo + r^-1*a_hi  equals  C + r^2 L + r^-2 R
    alo = [cref.limbs_to_int(x) for x in a[: n // 2]]
    ahi = [cref.limbs_to_int(x) for x in a[n // 2:]]
    a2 = P.scalars_to_array([(r * x + rinv * y) % C.order for x, y in zip(alo, ahi)], name)
    lhs = folded.commit(a2, is_mont=False)
    Cfull = gens.commit(a, is_mont=False)
    def scale(cm, k):
        aff = cm.to_affine().reshape(1, 8)
        return P.CommitmentGens(name, aff, precompute=False).commit(P.scalars_to_array([k], name), is_mont=False)
    rhs = Cfull + scale(L, r * r % C.order) + scale(Rr, rinv * rinv % C.order)
    assert lhs == rhs


@pytest.mark.parametrize("groups", [1, 0])
def test_ipa_without_generator_folding(groups, gpu_lib, cref):
    """reef_ipa_cross_terms == fold the generators k times (oracle), then the two cross MSMs."""
    from reef_amd import msm
    cid = 0
    C = CURVES["pallas"]
    n = 256
    gens0 = cref.gen_bases_ap(cid, 21, 4, n)
    rng = SplitMix64(77)
    with msm.MsmContext(cid, gens0, bucket_groups=groups) as ctx:
        gens = gens0
        w1s, w2s = [], []
        a = cref.gen_scalars(cid, 3, n)                       # Montgomery-form scalars
        for k in range(0, 6):
            n_k = n >> k
            half = n_k // 2
            a_k = a[:n_k].copy()
            L, R = ctx.ipa_cross_terms(a_k, w1s, w2s)
            exp_l = cref.msm_pippenger(cid, gens[half:n_k].copy(), a_k[:half].copy())
            exp_r = cref.msm_pippenger(cid, gens[:half].copy(), a_k[half:].copy())
            assert msm.compress(cid, L) == cref.compress(cid, exp_l), k
            assert msm.compress(cid, R) == cref.compress(cid, exp_r), k
            w1, w2 = uniform_scalar(rng, C.order), uniform_scalar(rng, C.order)
            gens = cref.fold(cid, np.ascontiguousarray(gens[:n_k]), w1, w2)
            w1s.append(w1)
            w2s.append(w2)


@pytest.mark.parametrize("byte_tables", [2, 1])
def test_ipa_cross_terms_at_prover_size_dlog_property(byte_tables, gpu_lib):
    """The L/R batch at the size of Reef's final IPA (2^16 generators: the batch goes through the two-level
    sort, or -- byte_tables = 1 -- both terms come from one pass over the byte tables).  Generators in arithmetic progression, so a cross term is (sum_j s[j] * (k0 + j*d)) * G with the
    expanded scalars of the definition: s_L[j] = a[i - n_k/2] * coef[t] on the upper half of block t,
    s_R[j] = a[i + n_k/2] * coef[t] on the lower half, coef[t] = prod_m (bit_{k-1-m}(t) ? w2_m : w1_m)."""
    from reef_amd import msm
    C = CURVES["pallas"]
    q = C.order
    n, k0, d = 1 << 16, 424242, 77
    gens = msm.gen_bases("pallas", k0, d, n)
    rng = SplitMix64(2025)
    a_canon = msm.gen_scalars("pallas", 31, n, mont=False)
    a_int = [sum(int(a_canon[i, j]) << (64 * j) for j in range(4)) for i in range(n)]
    with msm.MsmContext("pallas", gens, bucket_groups=1, byte_tables=byte_tables) as ctx:
        assert ctx.has_byte_tables() == (byte_tables == 1)
        w1s, w2s = [], []
        for k in range(0, 4):
            n_k, half = n >> k, n >> (k + 1)
            coef = []
            for t in range(1 << k):
                c = 1
                for m in range(k):
                    c = c * (w2s[m] if (t >> (k - 1 - m)) & 1 else w1s[m]) % q
                coef.append(c)
            accl = accr = 0
            for j in range(n):
                i, t = j & (n_k - 1), j // n_k
                if i >= half:
                    accl += a_int[i - half] * coef[t] % q * (k0 + j * d)
                else:
                    accr += a_int[i + half] * coef[t] % q * (k0 + j * d)
            L, R = ctx.ipa_cross_terms(np.ascontiguousarray(a_canon[:n_k]), w1s, w2s, is_mont=False)
            assert msm.compress("pallas", L) == C.compress(C.mul(accl % q, C.gen)), k
            assert msm.compress("pallas", R) == C.compress(C.mul(accr % q, C.gen)), k
            w1s.append(uniform_scalar(rng, q))
            w2s.append(uniform_scalar(rng, q))


def test_hyrax_bind_rows_is_consistent_with_commit(gpu_lib, cref):
    """prove_eval's row binding (commitment.rs:371-391): commit(LZ; LZ_blind) must equal
    sum_i L_i * C_i over the row commitments -- the relation the Hyrax verifier checks."""
    from oracle import mle_oracle
    from oracle.pasta_oracle import Q, SplitMix64
    from reef_amd import msm
    from reef_amd.provider import CommitmentGens, HyraxPC
    from reef_amd.sumcheck import array_to_ints, ints_to_array
    m, left, right = 9, 4, 5
    rows, cols = 1 << left, 1 << right
    gens = cref.gen_bases_ap(0, 3, 7, cols)
    h = cref.gen_bases_ap(0, 0xB11D, 1, 1)[0].copy()
    pc = HyraxPC(CommitmentGens("pallas", gens, h))
    z = cref.gen_scalars(0, 11, rows * cols, kind=2, small_bound=131)
    blinds = cref.gen_scalars(0, 12, rows)
    comms, comp = pc.commit(z, blinds)
    zsym = cref.gen_scalars(0, 11, rows * cols, kind=2, small_bound=131, mont=False)[:, 0].astype(np.uint8)
    assert bytes(pc.commit_symbols(zsym, blinds, 8)[1]) == bytes(comp)   # the same commitments from the document bytes
    rng = SplitMix64(3)
    R = 1 << 256
    point = [(rng.next() << 190 | rng.next()) % Q for _ in range(m)]
    pm = ints_to_array([v * R % Q for v in point])
    lz, ev, lz_blind = pc.bind_rows(z, blinds, pm)
    # left side: one commitment to LZ with blind LZ_blind
    lhs = cref.row_msm(0, gens, np.ascontiguousarray(lz), 1, cols, h=h, blinds=lz_blind.reshape(1, 4))
    # right side: sum_i L_i * C_i  (L in Montgomery form as scalars of an MSM over the row commitments)
    L = mle_oracle.eq_evals(point[:left], Q)
    aff = cref.to_affine(0, comms)
    rhs = cref.msm_naive(0, aff, ints_to_array([v * R % Q for v in L]))
    assert cref.compress(0, lhs) == cref.compress(0, rhs)
    zi = [v * pow(R, -1, Q) % Q for v in array_to_ints(z)]
    assert array_to_ints(ev)[0] == mle_oracle.evaluate(zi, point, Q) * R % Q


def test_cpp_provider_mirror_matches_oracle(gpu_lib, cref):
    """reef_amd/csrc/host/reef_provider.hpp (the C++ mirror of the provider interface Reef calls) through
    its self-test program: every value it prints is recomputed here with the oracle from the same seeds."""
    import json
    import os
    import subprocess
    from oracle import mle_oracle, sumcheck_oracle as S
    from reef_amd import _ffi
    exe = os.path.join(os.path.dirname(_ffi.LIB_PATH), "provider_selftest")
    assert os.path.exists(exe), "build() did not produce provider_selftest"
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stderr[-2000:]
    got = json.loads(run.stdout.strip().splitlines()[-1])
    # CE::commit
    n = 1000
    gens = cref.gen_bases_ap(0, 77, 13, n)
    h = cref.gen_bases_ap(0, 0xB11D, 1, 1)[0].copy()
    v = cref.gen_scalars(0, 4242, n, kind=1)
    blind = cref.gen_scalars(0, 4243, 1)
    assert got["commit"] == cref.compress(0, cref.msm_pippenger(0, gens, v, threads=4)).hex()
    assert got["commit_blind"] == cref.compress(0, cref.row_msm(0, gens, v, 1, n, h=h, blinds=blind)).hex()
    # HyraxPC::commit, both input forms
    rows, cols = 16, 32
    rg = cref.gen_bases_ap(0, 3, 7, cols)
    z = cref.gen_scalars(0, 11, rows * cols, kind=2, small_bound=131)
    bl = cref.gen_scalars(0, 12, rows)
    want = cref.compress(0, cref.row_msm(0, rg, z, rows, cols, h=h, blinds=bl))
    want_rows = [want[32 * i:32 * i + 32].hex() for i in range(rows)]
    assert got["hyrax"] == want_rows and got["hyrax_symbols"] == want_rows
    # the same through CommitmentGensOnDevices (device groups: three members on device 0)
    assert got["group_commit"] == got["commit"] and got["group_commit_blind"] == got["commit_blind"]
    assert got["group_commit_fanout"] == got["commit"] and got["group_timing_consistent"] is True      # round 6: REEF_SCALARS_FANOUT and the itemised call through the mirror
    assert got["group_hyrax"] == want_rows and got["group_hyrax_symbols"] == want_rows
    # prove_eval row binding
    zc = [int(x) for x in cref.gen_scalars(0, 11, rows * cols, kind=2, small_bound=131, mont=False)[:, 0]]
    point = [(0x1f83d9abfb41bd6b + j) | 0x5be0cd19137e2179 << 64 | 0x3c6ef372fe94f82b << 128 | 0x0a54ff53a5f1d36f << 192 for j in range(9)]
    lz, ev = mle_oracle.bound_rows(zc, point, 4, S.Q)
    assert got["bind_eval"] == ev.to_bytes(32, "little").hex() and got["bind_lz0"] == lz[0].to_bytes(32, "little").hex()
    # IPA cross terms over Vesta, rounds 0 and 2 (generators folded by the oracle)
    m = 256
    ig = cref.gen_bases_ap(1, 21, 4, m)
    a = cref.gen_scalars(1, 3, m)
    assert got["ipa_l0"] == cref.compress(1, cref.msm_pippenger(1, ig[m // 2:].copy(), a[:m // 2].copy())).hex()
    assert got["ipa_r0"] == cref.compress(1, cref.msm_pippenger(1, ig[:m // 2].copy(), a[m // 2:].copy())).hex()
    g2 = cref.fold(1, cref.fold(1, ig, 5, 11), 7, 13)
    q = m // 4
    assert got["ipa_l2"] == cref.compress(1, cref.msm_pippenger(1, g2[q // 2:].copy(), a[:q // 2].copy())).hex()
    assert got["ipa_r2"] == cref.compress(1, cref.msm_pippenger(1, g2[:q // 2].copy(), a[q // 2:q].copy())).hex()
    # the reference's mle_linear_basic inputs through the C++ SumCheck
    evals, qs, claims, last_q = [2, 3, 5, 7, 9, 13, 17, 19], [2, 1, 7], [3, 9, 27, 81], [5, 3, 2]
    t, e = list(evals), S.gen_eq_table(claims, qs, last_q)
    rs = [5, 1000003, 0x1234567890ABCDEF]
    for i in range(1, 4):
        want3 = [x.to_bytes(32, "little").hex() for x in S.linear_mle_coeffs(t, e, 3, i)]
        assert got["sumcheck"][i - 1] == want3, i
        S.linear_mle_fold(t, e, 3, i, rs[i - 1])
    assert got["sumcheck_final"] == t[0].to_bytes(32, "little").hex()
    assert got["error_throws"] is True
    # the same step through the C++ mirror's linear_mle_product: (r_i, xsq, x, con) per round, (con, x, xsq) absorbed in that order (r1cs_helper.rs:478-482)
    hx = lambda v: v.to_bytes(32, "little").hex()
    t, e = list(evals), S.gen_eq_table(claims, qs, last_q)
    for i in range(1, 4):
        xsq, x, con = S.linear_mle_coeffs(t, e, 3, i)
        assert got["lmp"][i - 1] == [hx(rs[i - 1]), hx(xsq), hx(x), hx(con)] and got["lmp_absorbed"][i - 1] == [hx(con), hx(x), hx(xsq)]
        S.linear_mle_fold(t, e, 3, i, rs[i - 1])
    assert got["lmp_final"] == hx(t[0])
    # CommitmentGens::new(label, n) / new_with_blinding_gen: generators from the label (stand-in constants = the oracle's), commitments over them
    from oracle import keygen_oracle as KO, merkle_oracle as MO
    from reef_amd import keygen
    kp = KO.standin_params("pallas")
    ln = 300
    raw = keygen.derive_generators("pallas", b"reef ck", ln, kp.a, kp.b, kp.z, kp.iso, kp.dst)
    assert keygen.points_to_ints("pallas", raw) == KO.from_label(b"reef ck", ln, kp)
    assert got["label_gens"] == raw[:3].tobytes().hex()
    lv = cref.gen_scalars(0, 99, ln, kind=1)
    assert got["label_commit"] == cref.compress(0, cref.msm_pippenger(0, raw, lv, threads=4)).hex()
    assert got["label_commit_blind"] == cref.compress(0, cref.row_msm(0, raw, lv, 1, ln, h=h, blinds=blind)).hex()
    # MerkleCommitment::new + path_wits on the reference's own document and on a longer one built in blocks
    pp = MO.standin_params()

    def path(doc, tree, q):
        return [[bool(lr), idx, opp.to_bytes(32, "little").hex()] for lr, idx, opp in MO.path_wits(doc, tree, q)]
    doc7 = [2, 3, 4, 5, 6, 7, 8]
    root7, tree7 = MO.commit(doc7, pp)
    assert got["merkle_root"] == root7.to_bytes(32, "little").hex() and got["merkle_levels"] == len(tree7)
    assert got["merkle_path6"] == path(doc7, tree7, 6) and got["merkle_path3"] == path(doc7, tree7, 3)
    long_doc = [(31 * i + 7) % 131 for i in range(1001)]
    rootl, treel = MO.commit(long_doc, pp)
    assert got["merkle_blocks_root"] == rootl.to_bytes(32, "little").hex() and got["merkle_blocks_path500"] == path(long_doc, treel, 500)
    assert got["merkle_oob_throws"] is True



@pytest.mark.parametrize("groups", [1, 0])
def test_lazy_fold_serves_the_whole_ipa(groups, gpu_lib, cref):
    """nova's ipa_pc as it is written -- split_at, two commits, fold, ... down to one generator -- on generators that RECORD
    their folds (FoldedGens over the resident key, reef_msm_folded): every L, R and the last generator equal what the
    oracle gets by folding the generators for real."""
    from reef_amd import provider as P
    cid, name = 0, "pallas"
    C = CURVES[name]
    n = 128
    root = P.CommitmentGens(name, cref.gen_bases_ap(cid, 33, 7, n), precompute=bool(groups))
    rng = SplitMix64(4242)
    lazy = P.FoldedGens(root)
    gens = root.bases                                           # the oracle's generators, folded for real every round
    a = cref.gen_scalars(cid, 9, n, mont=False)
    for k in range(7):
        n_k = n >> k
        half = n_k // 2
        assert len(lazy) == n_k
        g_lo, g_hi = lazy.split_at(half)
        L = g_hi.commit(a[:half].copy(), is_mont=False)
        R = g_lo.commit(a[half:n_k].copy(), is_mont=False)
        assert L.compress() == cref.compress(cid, cref.msm_pippenger(cid, np.ascontiguousarray(gens[half:n_k]), a[:half].copy(), mont=False)), k
        assert R.compress() == cref.compress(cid, cref.msm_pippenger(cid, np.ascontiguousarray(gens[:half]), a[half:n_k].copy(), mont=False)), k
        if k == 2:                                              # a ragged slice and the recorded folds performed for real
            assert (lazy.materialize().bases == gens[:n_k]).all()
            mid = FoldedGensSlice = P.FoldedGens(root, lazy.w1s, lazy.w2s, 3, 5)
            assert mid.commit(a[:5].copy(), is_mont=False).compress() == cref.compress(
                cid, cref.msm_pippenger(cid, np.ascontiguousarray(gens[3:8]), a[:5].copy(), mont=False))
        w1, w2 = uniform_scalar(rng, C.order), uniform_scalar(rng, C.order)
        lazy = lazy.fold(w1, w2)
        gens = cref.fold(cid, np.ascontiguousarray(gens[:n_k]), w1, w2)
        alo = [cref.limbs_to_int(x) for x in a[:half]]
        ahi = [cref.limbs_to_int(x) for x in a[half:n_k]]
        a = P.scalars_to_array([(w2 * x + w1 * y) % C.order for x, y in zip(alo, ahi)], name)   # any fold of a: only shapes matter here
    assert len(lazy) == 1
    one = P.scalars_to_array([1], name)
    last = lazy.commit(one, is_mont=False)                      # the last remaining generator, as a point
    assert last.to_affine().tobytes() == np.ascontiguousarray(gens[0]).tobytes()
    with pytest.raises(ValueError):
        lazy.fold(1, 2)
    with pytest.raises(ValueError):
        P.FoldedGens(root, [1], [2], 60, 10)
    root.close()


def test_ipa_cross_terms_on_byte_tables(gpu_lib, cref):
    """reef_ipa_cross_terms on a key with byte tables: both cross terms from ONE pass over the table entries (each point feeds
    exactly one of them) == fold the generators k times (oracle), then the two cross MSMs."""
    from reef_amd import msm
    cid = 0
    C = CURVES["pallas"]
    n = 2048
    gens0 = cref.gen_bases_ap(cid, 51, 3, n)
    rng = SplitMix64(1234)
    with msm.MsmContext(cid, gens0, bucket_groups=1, byte_tables=1) as ctx:
        assert ctx.has_byte_tables()
        gens = gens0
        w1s, w2s = [], []
        a = cref.gen_scalars(cid, 8, n)
        a[5] = 0
        for k in range(0, 9):
            n_k = n >> k
            half = n_k // 2
            a_k = a[:n_k].copy()
            L, R = ctx.ipa_cross_terms(a_k, w1s, w2s)
            exp_l = cref.msm_pippenger(cid, np.ascontiguousarray(gens[half:n_k]), a_k[:half].copy())
            exp_r = cref.msm_pippenger(cid, np.ascontiguousarray(gens[:half]), a_k[half:].copy())
            assert msm.compress(cid, L) == cref.compress(cid, exp_l), k
            assert msm.compress(cid, R) == cref.compress(cid, exp_r), k
            w1, w2 = uniform_scalar(rng, C.order), uniform_scalar(rng, C.order)
            gens = cref.fold(cid, np.ascontiguousarray(gens[:n_k]), w1, w2)
            w1s.append(w1)
            w2s.append(w2)


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0, 0]])
def test_commitment_gens_on_several_devices_give_the_same_points(devices, gpu_lib, cref):
    """CommitmentGens(..., devices=[..]): the provider mirror over the library's device groups (include/reef_msm.h section 5; the ordinals
    repeat device 0 on a one-GPU box) -- CE::commit with and without a blind (commitment.rs:350,361; framework.rs:668) and HyraxPC::commit from
    field elements and from document symbols (commitment.rs:187) return the points the one-device mirror and the oracle give."""
    from reef_amd import msm, provider as P
    cid, name = 0, "pallas"
    n = 3001
    bases = cref.gen_bases_ap(cid, 77, 5, n)
    H = cref.gen_bases_ap(cid, 0xB11D, 1, 1)[0]
    v = cref.gen_scalars(cid, 3, n, kind=1)
    b = cref.gen_scalars(cid, 4, 1)
    one = P.CommitmentGens(name, bases, H)
    for split in (msm.SPLIT_WINDOWS, msm.SPLIT_POINTS):
        many = P.CommitmentGens(name, bases, H, devices=devices, split=split)
        assert many.commit(v) == one.commit(v)
        assert many.commit(v).compress() == cref.compress(cid, cref.msm_pippenger(cid, bases, v, threads=4))
        if split == msm.SPLIT_WINDOWS:
            assert many.commit(v, b[0]) == one.commit(v, b[0])
            assert many.commit(v, b[0]).compress() == cref.compress(cid, cref.row_msm(cid, bases, v, 1, n, h=H, blinds=b, threads=4))
        else:
            with pytest.raises(ValueError):
                many.commit(v, b[0])
        many.close()
    num_vars = 13
    left, right = P.HyraxPC.compute_factored_lens(num_vars)
    rows, row_len = 1 << left, 1 << right
    gens_v = P.CommitmentGens(name, bases[:row_len].copy(), H, devices=devices)
    poly = cref.gen_scalars(cid, 7, rows * row_len, kind=2, small_bound=7)
    canon = cref.gen_scalars(cid, 7, rows * row_len, kind=2, small_bound=7, mont=False)
    blinds = cref.gen_scalars(cid, 8, rows)
    exp = cref.compress(cid, cref.row_msm(cid, gens_v.bases, poly, rows, row_len, h=H, blinds=blinds, threads=4))
    pc = P.HyraxPC(gens_v)
    assert pc.commit(poly, blinds)[1].tobytes() == exp
    assert pc.commit_symbols(canon[:, 0].astype(np.uint8), blinds, 3)[1].tobytes() == exp
    gens_v.close()
    one.close()
