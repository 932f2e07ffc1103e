"""world_size-2 gloo tests (CPU) of the multi-GPU path: point sharding, the all-gather of 96-byte
partial sums, the fixed combination order and the window split.  The group arithmetic is injected
(the oracle stands in for the GPU calls, which need a device); what is under test is
reef_amd/distributed.py -- the host logic the N > 1 bench path and a multi-GPU prover share."""
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
        cid = 0
        bases = R.gen_bases_ap(cid, 77, 3, n)
        scalars = R.gen_scalars(cid, 2024, n, kind=1)
        one = np.zeros((world, 4), dtype=np.uint64)
        one[:, 0] = 1

        def local_msm(b, s):
            return R.msm_pippenger(cid, np.ascontiguousarray(b), np.ascontiguousarray(s), threads=1)

        def add_points(jacs):
            aff = R.to_affine(cid, jacs)
            return R.msm_naive(cid, aff, one[: aff.shape[0]].copy(), mont=False)

        combined = D.sharded_msm(local_msm, add_points, bases, scalars)
        comp = R.compress(cid, combined)
        expect = R.compress(cid, R.msm_pippenger(cid, bases, scalars, threads=1))

        # window split: rank r owns windows w = r mod world (c = 16 -> 16 windows)
        c, W = 16, 16
        canon = np.zeros_like(scalars)
        for i in range(n):
            canon[i] = R.field_op("from_mont", 1, scalars[i].copy())   # Pallas scalars live in Fq

        def window_sums(ws):
            out = []
            for w in ws:
                dig = np.zeros((n, 4), dtype=np.uint64)
                word, sh = (w * c) // 64, (w * c) % 64
                dig[:, 0] = (canon[:, word] >> np.uint64(sh)) & np.uint64((1 << c) - 1)
                out.append(R.msm_pippenger(cid, bases, dig, mont=False, threads=1))
            return out

        def combine(sums):
            aff = R.to_affine(cid, np.stack(sums))
            sc = np.zeros((W, 4), dtype=np.uint64)
            for w in range(W):
                v = 1 << (c * w)
                for j in range(4):
                    sc[w, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
            return R.msm_naive(cid, aff, sc, mont=False)

        wcomp = R.compress(cid, D.window_sharded_msm(window_sums, combine, W))

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
        results[rank] = (comp == expect, wcomp == expect, comp.hex(), rows_ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [1001])
def test_sharded_msm_world2(n):
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, results), nprocs=world, join=True)
    assert len(results) == world
    assert all(results[r][0] for r in range(world)), "point-sharded result differs from the single-rank MSM"
    assert all(results[r][1] for r in range(world)), "window-sharded result differs from the single-rank MSM"
    assert results[0][2] == results[1][2], "ranks disagree on the combined point"
    assert all(results[r][3] for r in range(world)), "row-sharded commitments differ from the single-rank batch"
