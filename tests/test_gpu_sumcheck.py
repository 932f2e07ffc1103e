"""Row N2: the sum-check vector kernels (through the C ABI) against oracle/sumcheck_oracle.py,
which is pinned to the reference's own tests (tests/test_sumcheck_oracle.py).  Bit-exact:
every value is a canonical integer mod q."""
import numpy as np
import pytest

from oracle.pasta_oracle import SplitMix64, uniform_scalar
from oracle.sumcheck_oracle import Q, gen_eq_table, linear_mle_coeffs, linear_mle_fold, verifier_mle_eval

pytestmark = pytest.mark.gpu


def test_reference_test_vectors_on_gpu(gpu_lib):
    """mle_linear_basic (r1cs.rs:2411-2515) with the reference's inputs, run on the GPU."""
    from reef_amd.sumcheck import SumCheck
    evals = [2, 3, 5, 7, 9, 13, 17, 19]
    qs, last_q, claims = [2, 1, 7], [2, 3, 5], [3, 9, 27, 81]
    eq_ref = gen_eq_table(claims, qs, list(reversed(last_q)))
    with SumCheck("pallas", 3) as sc:
        sc.set_table(0, evals)
        sc.gen_eq_table(claims, qs, list(reversed(last_q)))
        assert sc.read(1, 8) == eq_ref
        t, e = list(evals), list(eq_ref)
        claim = sum(a * b for a, b in zip(t, e)) % Q
        rs = [5, Q - 3, 0x1234567890ABCDEF]
        for i in range(1, 4):
            got = sc.round_coeffs(i)
            assert got == linear_mle_coeffs(t, e, 3, i)
            xsq, x, con = got
            assert claim == (2 * con + x + xsq) % Q              # the reference's sanity assertion
            sc.fold(i, rs[i - 1])
            linear_mle_fold(t, e, 3, i, rs[i - 1])
            claim = (xsq * rs[i - 1] ** 2 + x * rs[i - 1] + con) % Q
            assert sc.read(0, 1 << (3 - i)) == t[: 1 << (3 - i)]
            assert sc.read(1, 1 << (3 - i)) == e[: 1 << (3 - i)]
        # prover_mle_partial_eval(table, sc_rs) == the folded table (r1cs.rs:2379-2385)
        assert sc.read(0, 1)[0] == verifier_mle_eval(evals, rs)
        sc.reset_table()                                         # next folding step starts from the table
        assert sc.read(0, 8) == evals


@pytest.mark.parametrize("curve,ell,n_t", [("pallas", 10, 1000), ("pallas", 14, 1 << 14), ("vesta", 9, 300)])
def test_full_sumcheck_vs_oracle(curve, ell, n_t, gpu_lib):
    """Random full-width table (zero-padded like the nldoc case), random eq inputs, every round."""
    from reef_amd.sumcheck import SumCheck
    from oracle.pasta_oracle import CURVES
    q = CURVES[curve].order
    rng = SplitMix64(ell * 1000 + n_t)
    table = [uniform_scalar(rng, q) for _ in range(n_t)]
    nq = 5
    qs = [rng.next() % (1 << ell) for _ in range(nq)]
    qs[1] = qs[0]                                                # repeated lookup index
    claim_r = uniform_scalar(rng, q)
    rs = [pow(claim_r, k + 1, q) for k in range(nq + 1)]
    last_q = [uniform_scalar(rng, q) for _ in range(ell)]
    t = table + [0] * ((1 << ell) - n_t)
    e = gen_eq_table(rs, qs, last_q, q)
    with SumCheck(curve, ell) as sc:
        sc.set_table(0, table)
        sc.gen_eq_table(rs, qs, last_q)
        assert sc.read(1, 1 << ell) == e
        challenges = []
        for i in range(1, ell + 1):
            assert sc.round_coeffs(i) == linear_mle_coeffs(t, e, ell, i, q), i
            r = uniform_scalar(rng, q)
            challenges.append(r)
            sc.fold(i, r)
            linear_mle_fold(t, e, ell, i, r, q)
        assert sc.read(0, 1) == [t[0]] and sc.read(1, 1) == [e[0]]
    if ell <= 10:
        assert t[0] == verifier_mle_eval(table + [0] * ((1 << ell) - n_t), challenges, q)


@pytest.mark.parametrize("ell,fused", [(21, False), (26, True)])
def test_baseline_size_identity(ell, fused, gpu_lib):
    """cfg3 size (1 MiB document -> 2^21 entries) and cfg4 size (16 MiB DNA, --hybrid: 2^26 entries,
    SURVEY.md 8): the sum-check identity claim = g(0) + g(1) round after round, the final claim
    T~(r) * EQ~(r), and an independent O(n) check of the first round's constant term.  Tables are
    generated on the device; document symbols < 131 / < 7 like the two configurations."""
    from reef_amd.sumcheck import SumCheck
    from reef_amd import msm
    n = 1 << ell
    bound = 131 if ell == 21 else 7
    d_doc = msm.gen_scalars("pallas", 0xD0C, n, kind=2, small_bound=bound, mont=False, device=True)   # canonical integers
    d_eq = msm.gen_scalars("pallas", 0xE9, n, kind=0, mont=False, device=True)
    half = n // 2
    # con = sum_{b < n/2} T[b] * EQ[b]: symbols (< 2^8) times 32-bit pieces of EQ, summed in chunks that cannot overflow 64 bits
    doc_lo = d_doc.to_host((half, 4))
    assert not doc_lo[:, 1:].any() and int(doc_lo[:, 0].max()) < bound
    sym = doc_lo[:, 0].copy()
    del doc_lo
    eq_lo = d_eq.to_host((half, 4))
    chunk = 1 << 20
    expect_con = 0
    for j in range(4):
        for piece, shift in ((eq_lo[:, j] & np.uint64(0xFFFFFFFF), 64 * j), (eq_lo[:, j] >> np.uint64(32), 64 * j + 32)):
            prod = sym * piece                                   # < 2^8 * 2^32
            expect_con += sum(int(prod[k:k + chunk].sum()) for k in range(0, half, chunk)) << shift
    del eq_lo, sym
    with SumCheck("pallas", ell) as sc:
        sc.set_table_device(0, d_doc.ptr, n)
        sc.set_table_device(1, d_eq.ptr, n)
        rng = SplitMix64(99)
        xsq, x, con = sc.round_coeffs(1)
        assert con == expect_con % Q
        claim = (2 * con + x + xsq) % Q
        for i in range(1, ell + 1):
            r = uniform_scalar(rng, Q)
            claim_next = (xsq * r * r + x * r + con) % Q
            if fused and i < ell:                                 # one pass per round (reef_sc_fold_and_next_coeffs)
                xsq, x, con = sc.fold_and_next_coeffs(i, r)
            else:
                sc.fold(i, r)
                if i < ell:
                    xsq, x, con = sc.round_coeffs(i + 1)
            claim = claim_next
            if i < ell:
                assert claim == (2 * con + x + xsq) % Q, i + 1
        t0, e0 = sc.read(0, 1)[0], sc.read(1, 1)[0]
        assert claim == t0 * e0 % Q                             # final claim = T~(r) * EQ~(r)
        with pytest.raises(Exception):
            sc.read(0, 2)                                       # only one live entry is left after the last fold
        sc.reset_table()                                        # the next folding step starts from the unfolded table
        assert sc.read(0, 4) == [int(v) for v in d_doc.to_host((4, 4))[:, 0]]


def test_folded_tables_reject_reads_and_rounds_beyond_their_live_entries(gpu_lib):
    """A fold writes entries [0, pow) only; what lies beyond is not the folded table and must not be served."""
    from reef_amd.sumcheck import SumCheck
    from reef_amd.msm import ReefError
    ell = 6
    t = list(range(1, 65))
    with SumCheck("pallas", ell) as sc:
        sc.set_table(0, t)
        sc.set_table(1, t)
        sc.fold(1, 3)                                            # 32 live entries per table
        assert len(sc.read(0, 32)) == 32 and len(sc.read(1, 32)) == 32
        for bad in (lambda: sc.read(0, 33), lambda: sc.read(1, 64), lambda: sc.round_coeffs(1), lambda: sc.fold(1, 5),
                    lambda: sc.fold_and_next_coeffs(1, 5)):
            with pytest.raises(ReefError):
                bad()
        sc.round_coeffs(2)                                       # the next round is fine
        sc.reset_table()                                         # T is whole again, EQ is still folded
        assert sc.read(0, 64) == t
        with pytest.raises(ReefError):
            sc.round_coeffs(1)
        sc.set_table(1, t)
        sc.round_coeffs(1)


def test_fused_fold_and_next_coeffs(gpu_lib):
    """The one-pass form (fold of round i + sums of round i+1) gives the same transcript."""
    from reef_amd.sumcheck import SumCheck
    ell = 12
    rng = SplitMix64(4321)
    t = [uniform_scalar(rng, Q) for _ in range(1 << ell)]
    e = [uniform_scalar(rng, Q) for _ in range(1 << ell)]
    with SumCheck("pallas", ell) as sc:
        sc.set_table(0, t)
        sc.set_table(1, e)
        g = sc.round_coeffs(1)
        for i in range(1, ell + 1):
            assert g == linear_mle_coeffs(t, e, ell, i), i
            r = uniform_scalar(rng, Q)
            linear_mle_fold(t, e, ell, i, r)
            if i < ell:
                g = sc.fold_and_next_coeffs(i, r)
            else:
                sc.fold(i, r)
        assert sc.read(0, 1) == [t[0]] and sc.read(1, 1) == [e[0]]


@pytest.mark.parametrize("curve,ell,env", [
    ("pallas", 12, {}),                                                              # shipped grids: split rounds of 1 .. 32 workgroups
    ("vesta", 11, {"REEF_SC_SPLIT_BLOCKS": "1"}),                                    # one workgroup walks all the items (16 passes at 2^10 pairs of pairs)
    ("pallas", 12, {"REEF_SC_SPLIT_BLOCKS": "3"}),                                   # workgroups with unequal numbers of passes, the last one ragged
    ("pallas", 12, {"REEF_SC_SPLIT_MAX": "0", "REEF_SC_ITEMS": "8", "REEF_SC_FLOOR": "2"}),   # no split; long threads down to two workgroups
    ("pallas", 12, {"REEF_SC_SPLIT_MAX": "64", "REEF_SC_ITEMS": "3", "REEF_SC_FLOOR": "1", "REEF_SC_BLOCKS": "5"}),   # grid-stride tails in the dense kernels
    ("pallas", 13, {"REEF_SC_ONE_LAUNCH": "0"}),                                     # every round through the second kernel (the split form is a one-launch form: off)
])
def test_small_and_mid_round_grids(curve, ell, env, gpu_lib, experiment_build, monkeypatch):
    """The round-4 grids of the dense rounds -- an item's four folds and three products on the four waves of a workgroup
    (k_sc_fold_coeffs_split), several pairs per thread in the mid-sized rounds, the wave sums through DPP, limb sums folded back by
    fe_from_limb_sums -- against the oracle round by round, with the edge challenges 0, 1 and q - 1 among the random ones and a
    second step on the same context."""
    from reef_amd.sumcheck import SumCheck
    from oracle.pasta_oracle import CURVES
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    q = CURVES[curve].order
    rng = SplitMix64(ell * 31 + len(env))
    n = 1 << ell
    with SumCheck(curve, ell) as sc:
        for step in range(2):
            t = [uniform_scalar(rng, q) for _ in range(n)]
            e = [uniform_scalar(rng, q) for _ in range(n)]
            t[3], t[n - 1], e[0], e[5] = 0, q - 1, q - 1, 0
            sc.set_table(0, t)
            sc.set_table(1, e)
            g = sc.round_coeffs(1)
            for i in range(1, ell + 1):
                assert g == linear_mle_coeffs(t, e, ell, i, q), (step, i)
                r = {3: 0, 5: 1, 7: q - 1}.get(i, uniform_scalar(rng, q))
                linear_mle_fold(t, e, ell, i, r, q)
                if i < ell:
                    g = sc.fold_and_next_coeffs(i, r)
                else:
                    sc.fold(i, r)
            assert sc.read(0, 1) == [t[0]] and sc.read(1, 1) == [e[0]]


@pytest.mark.parametrize("curve,ell,n_t,nq,fused", [("pallas", 2, 3, 1, False), ("pallas", 3, 8, 3, True), ("pallas", 7, 100, 9, True),
                                                     ("pallas", 11, 2000, 40, True), ("pallas", 12, 1 << 12, 33, False),
                                                     ("vesta", 9, 300, 5, True), ("pallas", 10, 1 << 10, 0, True)])
def test_rank_one_eq_rounds_vs_oracle(curve, ell, n_t, nq, fused, gpu_lib, experiment_build, monkeypatch):
    """gen_eq_table's table kept as FH (x) FL + point masses while the rounds fold FH bits (never written out): forced on at
    sizes the oracle handles (by default it serves tables of 2^22 entries and more), every round's coefficients, the folded
    tables, a look at EQ between rounds, repeated and colliding lookup indices, masses in both halves."""
    from reef_amd.sumcheck import SumCheck
    from oracle.pasta_oracle import CURVES
    monkeypatch.setenv("REEF_SC_RANK1_MIN_POW", "1")
    q = CURVES[curve].order
    rng = SplitMix64(ell * 77 + nq)
    table = [uniform_scalar(rng, q) for _ in range(n_t)]
    n = 1 << ell
    qs = [rng.next() % n for _ in range(nq)]
    if nq >= 3:
        qs[1] = qs[0]                                            # the same index twice
        qs[2] = qs[0] ^ (n >> 1)                                 # its partner in round 1: the masses meet after the first fold
    if nq >= 5:
        qs[3], qs[4] = 0, n - 1
    rs = [uniform_scalar(rng, q) for _ in range(nq + 1)]
    last_q = [uniform_scalar(rng, q) for _ in range(ell)]
    t = table + [0] * (n - n_t)
    e = gen_eq_table(rs, qs, last_q, q)
    with SumCheck(curve, ell) as sc:
        sc.set_table(0, table)
        for step in range(2):                                    # two folding steps on the same context (reset + a fresh EQ)
            tt, ee = list(t), list(e)
            sc.reset_table()
            sc.gen_eq_table(rs, qs, last_q)
            if step == 1:
                assert sc.read(1, n) == ee                       # written out for the reader; the rounds go on rank-one
            g = sc.round_coeffs(1)
            for i in range(1, ell + 1):
                assert g == linear_mle_coeffs(tt, ee, ell, i, q), (step, i)
                r = uniform_scalar(rng, q) if i != 2 else (0 if step == 0 else 1)     # the trivial challenges once
                linear_mle_fold(tt, ee, ell, i, r, q)
                live = 1 << (ell - i)
                if fused and i < ell:
                    g = sc.fold_and_next_coeffs(i, r)
                else:
                    sc.fold(i, r)
                    if i < ell:
                        g = sc.round_coeffs(i + 1)
                if i in (1, ell // 2, ell):
                    assert sc.read(0, live) == tt[:live], (step, i)
                    assert sc.read(1, live) == ee[:live], (step, i)
            assert sc.read(0, 1) == [tt[0]] and sc.read(1, 1) == [ee[0]]


def test_rank_one_matches_dense_at_default_size(gpu_lib, experiment_build, monkeypatch):
    """2^22 entries (the smallest table the rank-one rounds serve by default): the transcript of a whole folding step equals the
    dense form's (REEF_SC_RANK1=0), the sum-check identity holds round after round, and the final claim is T~(r) * EQ~(r)."""
    from reef_amd.sumcheck import SumCheck
    from reef_amd import msm
    ell = 22
    n = 1 << ell
    d_doc = msm.gen_scalars("pallas", 0xD0C, n, kind=2, small_bound=131, mont=False, device=True)
    rng = SplitMix64(2222)
    nq = 33
    rs = [uniform_scalar(rng, Q) for _ in range(nq + 1)]
    qs = [rng.next() % n for _ in range(nq)]
    last_q = [uniform_scalar(rng, Q) for _ in range(ell)]
    challenges = [uniform_scalar(rng, Q) for _ in range(ell)]
    transcripts = []
    with SumCheck("pallas", ell) as sc:
        sc.set_table_device(0, d_doc.ptr, n)
        for mode in ("1", "0"):
            monkeypatch.setenv("REEF_SC_RANK1", mode)
            sc.reset_table()
            sc.gen_eq_table(rs, qs, last_q)
            tr = []
            xsq, x, con = sc.round_coeffs(1)
            claim = (2 * con + x + xsq) % Q
            for i in range(1, ell + 1):
                tr.append((xsq, x, con))
                r = challenges[i - 1]
                claim = (xsq * r * r + x * r + con) % Q
                if i < ell:
                    xsq, x, con = sc.fold_and_next_coeffs(i, r)
                    assert claim == (2 * con + x + xsq) % Q, (mode, i + 1)
                else:
                    sc.fold(i, r)
            t0, e0 = sc.read(0, 1)[0], sc.read(1, 1)[0]
            assert claim == t0 * e0 % Q
            transcripts.append((tr, t0, e0))
    assert transcripts[0] == transcripts[1]


def _structured_table(kind, ell, rng, q):
    """Tables with the row structure of Reef's own (matrix view of 2^(ell//2) columns): a transition-table stub and one repeated
    fill value in the first half, symbols and zeros in the second (r1cs.rs:481-484, :2105-2112) -- and harder mixes."""
    n = 1 << ell
    cols = 1 << (ell // 2)
    rows = n // cols
    fill = uniform_scalar(rng, q)
    t = []
    for r in range(rows):
        if kind == "hybrid":
            if r == 0:
                row = [uniform_scalar(rng, q) for _ in range(cols)]                    # transitions: wide
            elif r < rows // 2:
                row = [fill] * cols                                                   # calc_fill up to the half
            elif r < rows - max(1, rows // 8):
                row = [rng.next() % 7 for _ in range(cols)]                           # symbols
            else:
                row = [0] * cols                                                      # padding
        elif kind == "document":
            row = [rng.next() % 131 for _ in range(cols)]
        else:                                                                         # "mixed": every class on both sides, edges of "small"
            pick = (r * 7 + 3) % 5
            if pick == 0:
                row = [uniform_scalar(rng, q) for _ in range(cols)]
            elif pick == 1:
                row = [(1 << 30) - 1 - (rng.next() % 3) for _ in range(cols)]         # the largest small entries
            elif pick == 2:
                row = [q - 1] * cols                                                  # constant, as large as it gets
            elif pick == 3:
                row = [5] * cols                                                      # constant and small: a K row
            else:
                row = [rng.next() % 4 for _ in range(cols)]
                if r % 2:
                    row[-1] = 1 << 30                                                 # one entry too large: the row is wide
        t += row
    return t


@pytest.mark.parametrize("kind,ell,nq,fused", [("hybrid", 10, 7, True), ("hybrid", 11, 40, True), ("document", 8, 3, True), ("mixed", 10, 9, True),
                                               ("mixed", 12, 33, False), ("hybrid", 6, 2, True), ("mixed", 7, 5, True)])
def test_structured_pristine_table_vs_oracle(kind, ell, nq, fused, gpu_lib, experiment_build, monkeypatch):
    """The first round of a folding step on a table with constant rows and rows of small entries (closed forms, 4-byte reads),
    forced on at sizes the oracle handles: coefficients of every round, the folded table after the first fold, two steps."""
    from reef_amd.sumcheck import SumCheck
    monkeypatch.setenv("REEF_SC_RANK1_MIN_POW", "1")
    q = Q
    rng = SplitMix64(ell * 131 + nq)
    t = _structured_table(kind, ell, rng, q)
    n = 1 << ell
    n_t = n if kind != "document" else n - 37                   # the last row of a document is cut short: zero padding
    t = t[:n_t] + [0] * (n - n_t)
    qs = [rng.next() % n for _ in range(nq)]
    qs[0] = 0
    qs[-1] = n - 1
    rs = [uniform_scalar(rng, q) for _ in range(nq + 1)]
    last_q = [uniform_scalar(rng, q) for _ in range(ell)]
    e = gen_eq_table(rs, qs, last_q, q)
    with SumCheck("pallas", ell) as sc:
        sc.set_table(0, t[:n_t])
        for step in range(2):
            tt, ee = list(t), list(e)
            sc.reset_table()
            sc.gen_eq_table(rs, qs, last_q)
            g = sc.round_coeffs(1)
            for i in range(1, ell + 1):
                assert g == linear_mle_coeffs(tt, ee, ell, i, q), (kind, step, i)
                r = uniform_scalar(rng, q) if not (step == 1 and i == 1) else q - 1
                linear_mle_fold(tt, ee, ell, i, r, q)
                live = 1 << (ell - i)
                if fused and i < ell:
                    g = sc.fold_and_next_coeffs(i, r)
                else:
                    sc.fold(i, r)
                    if i < ell:
                        g = sc.round_coeffs(i + 1)
                if i <= 2:
                    assert sc.read(0, live) == tt[:live], (kind, step, i)
            assert sc.read(0, 1) == [tt[0]] and sc.read(1, 1) == [ee[0]]
    # the same transcript without the structure
    monkeypatch.setenv("REEF_SC_STRUCT", "0")
    with SumCheck("pallas", ell) as sc:
        sc.set_table(0, t[:n_t])
        sc.gen_eq_table(rs, qs, last_q)
        assert sc.round_coeffs(1) == linear_mle_coeffs(t, e, ell, 1, q)


@pytest.mark.parametrize("kind,ell,nq", [("hybrid", 10, 7), ("hybrid", 12, 33), ("document", 8, 5), ("document", 11, 20), ("mixed", 10, 9), ("mixed", 13, 17),
                                         ("hybrid", 6, 2), ("hybrid", 7, 3)])
@pytest.mark.parametrize("interrupt", ["none", "read", "unfused", "coeffs", "set_eq"])
def test_deferred_first_fold(kind, ell, nq, interrupt, gpu_lib, either_build, monkeypatch):
    """Round 4: the first fold of a structured table is deferred by one round for the rows whose next fold has only constant / small
    sources too -- round two's sums come from the pristine rows with scaled row factors, the second fold writes a quarter-size table
    straight from the 4-byte sources.  Every round's coefficients against the oracle with NOTHING read in between (so that the
    second fold really takes the deferred path), masses planted in deferred rows, and the calls that are not a second fold -- a
    read of T, an unfused fold, stand-alone coefficients, a new EQ table -- right after the first fold: the table is written out
    after all and the step goes on as the reference's."""
    from reef_amd.sumcheck import SumCheck
    monkeypatch.setenv("REEF_SC_RANK1_MIN_POW", "1")
    q = Q
    rng = SplitMix64(ell * 7919 + nq + len(interrupt))
    t = _structured_table(kind, ell, rng, q)
    n = 1 << ell
    qs = [rng.next() % n for _ in range(nq)]
    qs[0], qs[-1] = n - 1, n // 2 + 1                           # masses in the (small / constant) second half
    rs = [uniform_scalar(rng, q) for _ in range(nq + 1)]
    last_q = [uniform_scalar(rng, q) for _ in range(ell)]
    e = gen_eq_table(rs, qs, last_q, q)
    for defer in ("1", "0"):
        monkeypatch.setenv("REEF_SC_DEFER", defer)
        with SumCheck("pallas", ell) as sc:
            sc.set_table(0, t)
            for step in range(2):
                tt, ee = list(t), list(e)
                sc.reset_table()
                sc.gen_eq_table(rs, qs, last_q)
                g = sc.round_coeffs(1)
                i = 1
                while i <= ell:
                    assert g == linear_mle_coeffs(tt, ee, ell, i, q), (kind, defer, step, i)
                    r = uniform_scalar(rng, q) if i != 2 else (q - 1, 0)[step]      # the second fold with the edge challenges
                    linear_mle_fold(tt, ee, ell, i, r, q)
                    live = 1 << (ell - i)
                    if i == 1 and step == 0 and interrupt == "unfused":
                        sc.fold(i, r)                                # not a fused call: nothing is deferred
                        g = sc.round_coeffs(i + 1)
                    elif i < ell:
                        g = sc.fold_and_next_coeffs(i, r)
                    else:
                        sc.fold(i, r)
                    if i == 1 and step == 0:
                        if interrupt == "read":
                            assert sc.read(0, live) == tt[:live], (kind, defer)
                        elif interrupt == "coeffs":
                            assert sc.round_coeffs(2) == linear_mle_coeffs(tt, ee, ell, 2, q)
                        elif interrupt == "set_eq" and ell > 1:
                            sc.set_table(1, ee[:live] + [0] * live)   # the folded EQ handed back as a dense table: same values
                            assert sc.round_coeffs(2) == linear_mle_coeffs(tt, ee, ell, 2, q)
                            g = sc.round_coeffs(2)
                    i += 1
                assert sc.read(0, 1) == [tt[0]] and sc.read(1, 1) == [ee[0]], (kind, defer, step)


@pytest.mark.parametrize("kind,ell,nq", [("hybrid", 10, 7), ("mixed", 9, 4)])
def test_gen_eq_before_set_table_gives_the_same_step(kind, ell, nq, gpu_lib, experiment_build, monkeypatch):
    """The results do not depend on the order of the calls (include/reef_msm.h 3b): gen_eq_table FIRST, then a structured table --
    and a structured table that replaces an unstructured one after gen_eq_table -- give the coefficients of the reference order
    (the sum of FL that constant rows need is taken in gen_eq_table whether or not a structure is known then: ADVICE r3)."""
    from reef_amd.sumcheck import SumCheck
    monkeypatch.setenv("REEF_SC_RANK1_MIN_POW", "1")
    q = Q
    rng = SplitMix64(ell * 977 + nq)
    t = _structured_table(kind, ell, rng, q)
    n = 1 << ell
    dense = [uniform_scalar(rng, q) for _ in range(n)]         # no structure: every row is W
    qs = [rng.next() % n for _ in range(nq)]
    rs = [uniform_scalar(rng, q) for _ in range(nq + 1)]
    last_q = [uniform_scalar(rng, q) for _ in range(ell)]
    e = gen_eq_table(rs, qs, last_q, q)
    for first in (None, dense):
        with SumCheck("pallas", ell) as sc:
            if first is not None:
                sc.set_table(0, first)                          # the table present at gen_eq time is not the one the rounds run on
            sc.gen_eq_table(rs, qs, last_q)
            sc.set_table(0, t)
            tt, ee = list(t), list(e)
            g = sc.round_coeffs(1)
            for i in range(1, ell + 1):
                assert g == linear_mle_coeffs(tt, ee, ell, i, q), (kind, first is None, i)
                r = uniform_scalar(rng, q)
                linear_mle_fold(tt, ee, ell, i, r, q)
                if i < ell:
                    g = sc.fold_and_next_coeffs(i, r)
                else:
                    sc.fold(i, r)
            assert sc.read(0, 1) == [tt[0]] and sc.read(1, 1) == [ee[0]]


def test_structured_table_at_cfg4_shape(gpu_lib, experiment_build, monkeypatch):
    """2^24 entries shaped like the hybrid table of BASELINE's cfg4 (first half: a few transition rows, then one value; second
    half: DNA symbols, then zeros): the transcript of a folding step equals the one taken without the structure and the dense one."""
    from reef_amd.sumcheck import SumCheck
    ell = 24
    n = 1 << ell
    cols = 1 << (ell // 2)
    rng = SplitMix64(2424)
    fill = uniform_scalar(rng, Q)
    arr = np.zeros((n, 4), dtype=np.uint64)
    head = 3 * cols + 17                                        # transitions spill into a fourth row
    for i in range(head):
        v = uniform_scalar(rng, Q)
        arr[i] = [(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    arr[head:n // 2] = [(fill >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)]
    doc_len = n // 2 - 12345
    arr[n // 2:n // 2 + doc_len, 0] = np.random.default_rng(5).integers(0, 7, size=doc_len, dtype=np.uint64)
    nq = 33
    rs = [uniform_scalar(rng, Q) for _ in range(nq + 1)]
    qs = [rng.next() % n for _ in range(nq)]
    last_q = [uniform_scalar(rng, Q) for _ in range(ell)]
    challenges = [uniform_scalar(rng, Q) for _ in range(ell)]
    from reef_amd import msm
    d_tab = msm.DeviceBuffer.from_host(arr)
    transcripts = []
    for env in ({}, {"REEF_SC_STRUCT": "0"}, {"REEF_SC_RANK1": "0"}):
        for k in ("REEF_SC_STRUCT", "REEF_SC_RANK1"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with SumCheck("pallas", ell) as sc:
            sc.set_table_device(0, d_tab.ptr, n)
            sc.gen_eq_table(rs, qs, last_q)
            tr = []
            g = sc.round_coeffs(1)
            for i in range(1, ell + 1):
                tr.append(g)
                if i < ell:
                    g = sc.fold_and_next_coeffs(i, challenges[i - 1])
                else:
                    sc.fold(i, challenges[i - 1])
            transcripts.append((tr, sc.read(0, 1)[0], sc.read(1, 1)[0]))
    assert transcripts[0] == transcripts[1] == transcripts[2]


# ---------------------------------------------------------------- the one-launch rounds under load ----
_STRESS_GRIDS = {
    # name: (ell, environment) -- REEF_SC_ITEMS=1 gives a round ceil(pairs / 256) workgroups up to REEF_SC_BLOCKS, so a step of 2^ell entries
    # walks through every power of two below the cap; REEF_SC_SPLIT_MAX=0 keeps the dense kernels on the small rounds
    "2-blocks": (12, {"REEF_SC_BLOCKS": "2", "REEF_SC_ITEMS": "1", "REEF_SC_SPLIT_MAX": "0"}),
    "3-blocks": (18, {"REEF_SC_BLOCKS": "3", "REEF_SC_ITEMS": "1", "REEF_SC_SPLIT_MAX": "0"}),
    "16-blocks": (16, {"REEF_SC_BLOCKS": "16", "REEF_SC_ITEMS": "1", "REEF_SC_SPLIT_MAX": "0"}),
    "64-blocks": (18, {"REEF_SC_BLOCKS": "64", "REEF_SC_ITEMS": "1", "REEF_SC_SPLIT_MAX": "0"}),
    "up-to-1024-blocks": (20, {"REEF_SC_BLOCKS": "2048", "REEF_SC_ITEMS": "1", "REEF_SC_SPLIT_MAX": "0", "REEF_SC_ONE_LAUNCH_MAX": "8192"}),   # every XCD many times over
    "split-rounds-64-blocks": (17, {}),                                                                                                         # the shipped grid: four-wave items, up to 64 workgroups
    "split-rounds-1024-blocks": (18, {"REEF_SC_SPLIT_BLOCKS": "1024", "REEF_SC_SPLIT_MAX": "65536"}),
    "rank-one-accumulator-sets": (16, {"REEF_SC_RANK1_MIN_POW": "1"}),                                                                          # k_sc_r1_final's 16 sets (a kernel boundary orders those)
}


def _stress(name, steps, order, load=True, extra=()):
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ell, env = _STRESS_GRIDS[name]
    e = dict(os.environ, REEF_SC_FENCE=str(order), **env)
    out = subprocess.run([os.path.join(root, "reef_amd", "_lib", "sc_stress"), str(ell), str(steps)] + (["load"] if load else []) + list(extra),
                         capture_output=True, text=True, timeout=900, env=e)
    assert out.returncode in (0, 1), out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["ell"] == ell and line["steps"] == steps and line["REEF_SC_FENCE"] == str(order)
    return line


@pytest.mark.parametrize("name,steps", [("2-blocks", 6000), ("3-blocks", 5000), ("16-blocks", 5000), ("64-blocks", 5000), ("up-to-1024-blocks", 3000),
                                        ("split-rounds-64-blocks", 5000), ("split-rounds-1024-blocks", 3000), ("rank-one-accumulator-sets", 3000)])
def test_one_launch_rounds_under_load(name, steps, gpu_lib):
    """The hand-over of a multi-block one-launch round (sc_round_epilogue) in the SHIPPED form -- REEF_SC_FENCE=2, the acq_rel ticket the HIP
    memory model blesses (default since round 6) -- repeated on the same inputs while two other caller threads run k_accum0 and stream a
    2^24-entry table: ~5.8 x 10^5 rounds in all, every coefficient triple against the two-launch form of the same step (a kernel boundary
    instead of the ticket)."""
    line = _stress(name, steps, 2)
    assert line["mismatched_steps"] == 0 and line["mismatched_values"] == 0, line
    assert line["load_msms"] > 0 and line["load_streaming_rounds"] > 0, line          # the other streams really ran beside it


@pytest.mark.parametrize("order,steps", [(0, 2500), (1, 600)])
@pytest.mark.parametrize("name", ["3-blocks", "64-blocks", "split-rounds-64-blocks"])
def test_one_launch_rounds_other_forms(name, order, steps, gpu_lib):
    """REEF_SC_FENCE=0 (the fence-free hand-over rounds 4-5 shipped: returning atomics + s_waitcnt; now the documented opt-in, 24 us per
    cfg3 step faster, resting on where gfx950 performs sc1 atomics) and =1 (__threadfence before the ticket) stay built and tested."""
    line = _stress(name, steps, order)
    assert line["mismatched_steps"] == 0, line


def test_stress_reference_transcript_is_the_oracles(gpu_lib):
    """What sc_stress compares against is itself checked: its two-launch reference transcript for a 2^12-entry table equals
    oracle/sumcheck_oracle.py's on the same inputs (table from the library's seeded generator, the inputs of make_inputs())."""
    from reef_amd import msm
    from reef_amd.sumcheck import array_to_ints
    ell = 12
    line = _stress("2-blocks", 2, 0, load=False, extra=("print",))
    got = [int(h, 16) for h in line["transcript"]]
    table = array_to_ints(msm.gen_scalars("pallas", 0x7AB1E + ell, 1 << ell, kind=0, mont=False))
    limbs = lambda l0, l1, l2, l3: l0 | (l1 << 64) | (l2 << 128) | (l3 << 192)
    nq = 9
    rs = [limbs(0x9e3779b97f4a7c15 * (i + 1) & (2**64 - 1), 0x1234 + i, 0x55aa * i, 0x0123456789abcdef) for i in range(nq + 1)]
    qs = [((0x2545F4914F6CDD1D * (i + 7)) & (2**64 - 1)) >> (64 - ell) for i in range(nq)]
    qs[1] = qs[0]
    last_q = [limbs(0xabcdef12345 + j, 0x77 * j, 0xfeed, 0x0fedcba987654321) for j in range(ell)]
    chal = [limbs(0x5851f42d4c957f2d + i, 0x14057b7ef767814f, 0x0123456789abcdef, 0x0fedcba987654321) for i in range(1, ell + 1)]
    assert all(v < Q for v in rs + last_q + chal)
    t, e = list(table), gen_eq_table(rs, qs, last_q)
    want = []
    for i in range(1, ell + 1):
        want += list(linear_mle_coeffs(t, e, ell, i))
        linear_mle_fold(t, e, ell, i, chal[i - 1])
    want.append(t[0])
    assert got == want


@pytest.mark.parametrize("ell,n_t", [(3, 8), (11, 1500)])
def test_linear_mle_product_drives_a_step_like_the_reference(ell, n_t, gpu_lib):
    """The product-side SumCheck.linear_mle_product(i, sponge) is the reference's one-call round (r1cs_helper.rs:441-506): sums, absorb
    (con, x, xsq), squeeze r_i, fold.  A whole step driven as r1cs.rs:2318-2385 drives it, with the same sponge (oracle/sumcheck_oracle.py:
    Sponge, the SpongeAPI calls Reef makes) on both sides: the transcripts (r_i, xsq, x, con) are the oracle's round for round, twice in a row
    (a second folding step after reset_table), and the final value is verifier_mle_eval at the challenges."""
    from oracle import merkle_oracle as MO
    from oracle.sumcheck_oracle import Sponge, linear_mle_product
    from reef_amd.sumcheck import SumCheck
    rng = SplitMix64(ell * 1000 + n_t)
    table = [2, 3, 5, 7, 9, 13, 17, 19] if ell == 3 else [uniform_scalar(rng, Q) for _ in range(n_t)]
    pp = MO.standin_params()
    with SumCheck("pallas", ell) as sc:
        sc.set_table(0, table)
        for step in range(2):
            nq = 3 + step
            claims = [uniform_scalar(rng, Q) for _ in range(nq + 1)]
            qs = [rng.next() % (1 << ell) for _ in range(nq)]
            last_q = [uniform_scalar(rng, Q) for _ in range(ell)]
            if step:
                sc.reset_table()
            sc.gen_eq_table(claims, qs, last_q)
            t = list(table) + [0] * ((1 << ell) - len(table))
            e = gen_eq_table(claims, qs, last_q)
            sp_gpu, sp_ref = Sponge(pp, 0x5EED + step), Sponge(pp, 0x5EED + step)
            sp_gpu.absorb(claims)
            sp_ref.absorb(claims)                               # Absorb(k) of the IO pattern, before the rounds (r1cs.rs:2260-2284)
            rs = []
            for i in range(1, ell + 1):
                got = sc.linear_mle_product(i, sp_gpu)
                assert got == linear_mle_product(t, e, ell, i, sp_ref), (step, i)
                rs.append(got[0])
            assert sc.read(0, 1)[0] == t[0] == verifier_mle_eval(list(table) + [0] * ((1 << ell) - len(table)), rs)
            assert sp_gpu.state == sp_ref.state
