"""world_size-2 gloo tests (CPU) of the multi-GPU path: point sharding, the window split with the engine's
semantics (partial sums that add up), the all-gather of 96-byte partial sums and the fixed combination
order.  The group arithmetic is injected (the oracle stands in for the GPU calls, which need a device);
what is under test is reef_amd/distributed.py -- PartialSumExchange and the split functions that
`bench.py --gpus N` drives with the C-ABI calls (tests/test_gpu_multirank.py runs that with two real
ranks on a GPU)."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from reef_amd import distributed as D


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [D.window_owner(w, 4) for w in range(6)] == [0, 1, 2, 3, 0, 1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pasta_ref as R
        from oracle.pasta_oracle import CURVES
        cid = 0
        order = CURVES["pallas"].order
        bases = R.gen_bases_ap(cid, 77, 3, n)
        scalars = R.gen_scalars(cid, 2024, n, kind=1)
        one = np.zeros((world, 4), dtype=np.uint64)
        one[:, 0] = 1

        def local_msm(b, s):
            return R.msm_pippenger(cid, np.ascontiguousarray(b), np.ascontiguousarray(s), threads=1)

        def sum_points(jacs):                                    # stand-in for reef_msm_ctx_sum_points
            aff = R.to_affine(cid, jacs)
            return R.msm_naive(cid, aff, one[: aff.shape[0]].copy(), mont=False)

        combined = D.point_sharded_msm(local_msm, sum_points, bases, scalars)
        comp = R.compress(cid, combined)
        expect = R.compress(cid, R.msm_pippenger(cid, bases, scalars, threads=1))

        # window split as the engine does it (reef_msm_ctx_set_window_split): every rank recodes the whole scalar
        # into signed c-bit digits and keeps the windows w = rank (mod world); its partial sum already carries the
        # weights 2^(c*w), so partials combine by plain addition, exactly like point-sharded ones
        c, W = 13, 20
        canon = [R.limbs_to_int(R.field_op("from_mont", 1, scalars[i].copy())) for i in range(n)]   # Pallas scalars live in Fq
        partial_holder = {}

        def partial_msm(r, wd):
            mine = D.owned_windows(W, wd, r)
            sc = np.zeros((n, 4), dtype=np.uint64)
            for i, k in enumerate(canon):
                dig = D.signed_digits(k, c, W)
                v = sum(dig[w] << (c * w) for w in mine) % order
                sc[i] = [(v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)]
            partial_holder["p"] = R.msm_pippenger(cid, bases, sc, mont=False, threads=1)
            return partial_holder["p"]

        wcomp = R.compress(cid, D.window_split_msm(partial_msm, sum_points))
        partial_is_not_total = R.compress(cid, partial_holder["p"]) != expect

        # row sharding (Hyrax commit): 5 rows over the same 64 generators, dealt out as 3 + 2
        rows, row_len = 5, 64
        rb = bases[:row_len].copy()
        rs = R.gen_scalars(cid, 7, rows * row_len, kind=2, small_bound=131)
        bl = R.gen_scalars(cid, 8, rows)
        h = R.gen_bases_ap(cid, 0xB11D, 1, 1)[0].copy()

        def local_rows(lo, hi):
            return R.row_msm(cid, rb, rs[lo * row_len:hi * row_len].copy(), hi - lo, row_len, h=h, blinds=bl[lo:hi].copy())

        got_rows = D.sharded_rows(local_rows, rows, 12)
        want_rows = R.row_msm(cid, rb, rs, rows, row_len, h=h, blinds=bl)
        rows_ok = R.compress(cid, got_rows) == R.compress(cid, want_rows)
        results[rank] = (comp == expect, wcomp == expect, comp.hex(), rows_ok, partial_is_not_total, wcomp.hex())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [301])
def test_sharded_msm_world2(n):
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, results), nprocs=world, join=True)
    assert len(results) == world
    assert all(results[r][0] for r in range(world)), "point-sharded result differs from the single-rank MSM"
    assert all(results[r][1] for r in range(world)), "window-split result differs from the single-rank MSM"
    assert results[0][2] == results[1][2] and results[0][5] == results[1][5], "ranks disagree on the combined point"
    assert all(results[r][3] for r in range(world)), "row-sharded commitments differ from the single-rank batch"
    assert all(results[r][4] for r in range(world)), "a window-split partial equals the whole MSM: nothing was split"


def test_signed_digits_recompose_and_split():
    """Host restatement of the engine's digit recoding: digits recompose the scalar, stay in the signed
    range, and the windows dealt out to the ranks partition them."""
    from oracle.pasta_oracle import CURVES, SplitMix64, uniform_scalar
    q = CURVES["pallas"].order
    rng = SplitMix64(17)
    for c in (4, 13, 16, 17):
        W = -(-256 // c)
        for k in [0, 1, q - 1, (1 << (c - 1)), (1 << (c - 1)) + 1, (1 << c) - 1] + [uniform_scalar(rng, q) for _ in range(50)]:
            d = D.signed_digits(k, c, W)
            assert sum(v << (c * w) for w, v in enumerate(d)) == k
            assert all(-(1 << (c - 1)) < v <= (1 << (c - 1)) for v in d)
        for world in (1, 2, 3, 8):
            owned = [D.owned_windows(W, world, r) for r in range(world)]
            assert sorted(w for o in owned for w in o) == list(range(W))


# ---- round 4: independent units over ranks, and the sum-check table sharded by low index bits -------------------------------
class _OracleSumCheck:
    """The oracle behind the method names of reef_amd.sumcheck.SumCheck (what LowBitShardedSumCheck drives on the GPU)."""

    def __init__(self, ell):
        from oracle import sumcheck_oracle as S
        self.S, self.ell, self.t, self.e = S, ell, None, None

    def set_table(self, which, values):
        v = list(values) + [0] * ((1 << self.ell) - len(values))
        if which == 0:
            self.t = v
        else:
            self.e = v

    def gen_eq_table(self, rs, qs, last_q):
        self.e = self.S.gen_eq_table(list(rs), list(qs), list(last_q)) if self.ell else [rs[-1] % self.S.Q]

    def round_coeffs(self, i):
        return self.S.linear_mle_coeffs(self.t, self.e, self.ell, i)

    def fold(self, i, r):
        self.S.linear_mle_fold(self.t, self.e, self.ell, i, r)

    def fold_and_next_coeffs(self, i, r):
        self.fold(i, r)
        return self.round_coeffs(i + 1)

    def read(self, which, count):
        return (self.t if which == 0 else self.e)[:count]


def _challenge(i, xsq, x, con):
    import hashlib
    from oracle.sumcheck_oracle import Q
    h = hashlib.sha256(b"%d|%x|%x|%x" % (i, con, x, xsq)).digest()          # absorbed in the reference's order (con, x, xsq)
    return int.from_bytes(h, "little") % Q


def _worker_units(rank, world, port, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import sumcheck_oracle as S
        from oracle.pasta_oracle import SplitMix64, uniform_scalar
        # (1) three independent units of different sizes and costs: every rank ends up with all results, each computed once
        widths, costs = [24 * 5, 24 * 4, 24 * 2], [5.4, 4.4, 2.9]
        ran = []

        def run_unit(u):
            ran.append(u)
            return np.arange(widths[u], dtype=np.uint64) * (u + 1) + 1000 * u
        got = D.run_placed_units(run_unit, widths, costs)
        units_ok = all((got[u] == np.arange(widths[u], dtype=np.uint64) * (u + 1) + 1000 * u).all() for u in range(3))
        owner = D.place_units(costs, world)
        units_ok = units_ok and sorted(ran) == [u for u in range(3) if owner[u] == rank]
        # (2) one folding step of the sum-check with the table sharded by low index bits against the single-rank transcript
        ell, nq = 7, 5
        rng = SplitMix64(99)
        table = [uniform_scalar(rng, S.Q) if i % 3 else i % 7 for i in range((1 << ell) - 9)]      # ragged: zero padding at the end
        qs = [rng.next() % (1 << ell) for _ in range(nq)]
        qs[0], qs[1] = 1, (1 << ell) - 2                                                             # one mass on each rank
        rs = [uniform_scalar(rng, S.Q) for _ in range(nq + 1)]
        last_q = [uniform_scalar(rng, S.Q) for _ in range(ell)]
        sh = D.LowBitShardedSumCheck(_OracleSumCheck(ell - (world.bit_length() - 1)), ell, S.Q)
        sh.set_table(table)
        sh.gen_eq_table(rs, qs, last_q)
        coeffs, chal, t_fin, e_fin = sh.run_step(_challenge)
        tt = table + [0] * ((1 << ell) - len(table))
        ee = S.gen_eq_table(rs, qs, last_q)
        want = []
        for i in range(1, ell + 1):
            c3 = S.linear_mle_coeffs(tt, ee, ell, i)
            want.append(c3)
            S.linear_mle_fold(tt, ee, ell, i, _challenge(i, *c3))
        results[rank] = (units_ok, coeffs == want, (t_fin, e_fin) == (tt[0], ee[0]), chal == [_challenge(i + 1, *c) for i, c in enumerate(want)])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_independent_units_and_low_bit_sharded_sumcheck(world):
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker_units, args=(world, _free_port(), results), nprocs=world, join=True)
    assert len(results) == world
    for r in range(world):
        assert results[r][0], "placed units: a result is wrong or a unit ran on the wrong rank"
        assert results[r][1], "sharded sum-check: round coefficients differ from the single-rank transcript"
        assert results[r][2], "sharded sum-check: final T~(r), EQ~(r) differ"
        assert results[r][3], "sharded sum-check: challenges differ"


def test_place_units_is_deterministic_and_balanced():
    assert D.place_units([5.4, 4.4, 2.9], 1) == [0, 0, 0]
    assert D.place_units([5.4, 4.4, 2.9], 2) == [0, 1, 1]          # the final SNARK's three arguments (cfg3, ms) on two GPUs: 5.4 | 7.3
    assert D.place_units([5.4, 4.4, 2.9], 3) == [0, 1, 2] == D.place_units([5.4, 4.4, 2.9], 8)
    assert D.place_units([], 4) == []
