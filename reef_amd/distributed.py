"""One-process-per-GPU sharding of the MSM path: the host logic `bench.py --gpus N` and a
multi-GPU prover share.

Every split ends in the same exchange: each rank holds one 96-byte PARTIAL sum (a Jacobian point), the N
partials are all-gathered and added in rank order on every rank (RCCL has no elliptic-curve reduction, so the
"reduce of partial bucket sums" of north_star is an all-gather followed by N-1 on-device additions;
payloads are 96 B per rank, so latency, not the 7 x 153 GB/s xGMI links, is what it costs).

  * by POINTS (`point_sharded_msm`): rank r keeps pairs [r*n/N, (r+1)*n/N) -- its slice of the commitment
    key stays resident on its GPU, its slice of the scalars is the only per-call upload -- and computes a
    full MSM over them.  Weak scaling in `bench.py` (2^20 pairs per GPU).
  * by WINDOW (`window_split_msm`, the split north_star names): every rank keeps ALL points and receives
    ALL scalars, recodes them in full (carries cross window borders) and accumulates only the Pippenger
    windows w = rank (mod N) (`reef_msm_ctx_set_window_split`).  With a pre-shifted key the weight 2^(c*w)
    is already in the table a digit reads, so a rank's result is a partial SUM and the combination is the
    same plain addition as for points -- no Horner step across ranks.  Strong scaling in `bench.py`
    (`--sharding windows`): one 2^20-point MSM, 16/N windows per GPU.
  * independent units (`sharded_rows`): the rows of a Hyrax commitment share the row generators
    (replicated, resident) and are dealt out in contiguous blocks; one all-gather of the commitments.

The arithmetic is injected (`partial_msm`, `sum_points`): on the GPU these are C-ABI calls of reef_amd.msm
(and the exchange runs on device tensors, ordered on the MSM's own HIP stream); the world_size-2 gloo tests
on CPU inject the oracle as a stand-in -- including an oracle version of the engine's digit recoding for
the window split -- and drive the SAME functions.  There is no CPU fallback in the product path.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

POINT_BYTES = 96


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced point ranges; the first n % world ranks get one extra pair."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def window_owner(window: int, world: int) -> int:
    return window % world


def owned_windows(n_windows: int, world: int, rank: int) -> List[int]:
    """The windows rank `rank` accumulates under the window split (reef_msm_ctx_set_window_split)."""
    return [w for w in range(n_windows) if window_owner(w, world) == rank]


def signed_digits(k: int, c: int, n_windows: int) -> List[int]:
    """Signed c-bit digits of a canonical scalar, as the engine's recoder emits them (k_recode): a raw window
    value above 2^(c-1) becomes raw - 2^c with a carry into the next window, so d_w in [-(2^(c-1)-1), 2^(c-1)]
    and sum_w d_w * 2^(c*w) = k.  Host restatement used by the CPU tests of the window split."""
    half, out, carry = 1 << (c - 1), [], 0
    for w in range(n_windows):
        raw = ((k >> (w * c)) & ((1 << c) - 1)) + carry
        if raw > half:
            out.append(raw - (1 << c))
            carry = 1
        else:
            out.append(raw)
            carry = 0
    return out


class PartialSumExchange:
    """All-gather of one 96-byte partial sum per rank, then the sum of the N points in rank order.

    `part`, `gathered` and `result` are torch uint8 tensors of 96, 96*N and 96 bytes that live wherever the
    MSM writes its result: on the GPU with backend "nccl" (= RCCL; the collective can be ordered on the
    MSM's own HIP stream through `stream_ctx`, so a step needs no host sync) or, for the host-staged
    fallback and for the CPU tests, anywhere with backend "gloo" (96 B per rank go through host memory).
    `sum_points(gathered, n, result)` is the injected addition (reef_msm_ctx_sum_points on the GPU)."""

    def __init__(self, sum_points: Callable, *, backend: str, group=None, before_exchange: Optional[Callable[[], None]] = None,
                 stream_ctx: Optional[Callable] = None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = backend
        self.sum_points = sum_points
        self.before_exchange = before_exchange     # e.g. ctx.sync when the collective is NOT on the MSM's stream
        self.stream_ctx = stream_ctx               # context manager factory putting the collective on the MSM's stream

    def all_gather(self, part, gathered) -> None:
        import torch
        dist = self.dist
        if self.backend == "nccl":
            if self.stream_ctx is not None:
                with self.stream_ctx():             # ordered after the MSM on the same stream: no host sync
                    dist.all_gather_into_tensor(gathered, part, group=self.group)
            else:
                if self.before_exchange:
                    self.before_exchange()
                dist.all_gather_into_tensor(gathered, part, group=self.group)
                torch.cuda.current_stream().synchronize()
        else:                                       # host-staged: 96 B per rank through host memory
            if self.before_exchange:
                self.before_exchange()
            host = part.cpu()
            outs = [torch.empty_like(host) for _ in range(self.world)]
            dist.all_gather(outs, host, group=self.group)
            gathered.copy_(torch.cat(outs))
            if gathered.is_cuda:
                torch.cuda.synchronize()

    def combine(self, part, gathered, result):
        """-> result = sum of every rank's `part`, bit-identical on all ranks (fixed order 0..N-1)."""
        self.all_gather(part, gathered)
        self.sum_points(gathered, self.world, result)
        return result


def _host_exchange(sum_points_np: Callable[[np.ndarray], np.ndarray], group=None):
    """PartialSumExchange over host tensors for numpy callers (CPU tests, simple drivers)."""
    import torch

    def adapter(gathered, n, result):
        pts = gathered.numpy().view(np.uint64).reshape(n, 12)
        result.copy_(torch.from_numpy(np.ascontiguousarray(sum_points_np(pts), dtype=np.uint64).view(np.uint8).copy()))

    return PartialSumExchange(adapter, backend="gloo", group=group)


def _combine_np(partial: np.ndarray, sum_points_np: Callable[[np.ndarray], np.ndarray], group=None) -> np.ndarray:
    import torch
    ex = _host_exchange(sum_points_np, group)
    part = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.uint64).reshape(12).view(np.uint8).copy())
    gathered = torch.zeros(POINT_BYTES * ex.world, dtype=torch.uint8)
    result = torch.zeros(POINT_BYTES, dtype=torch.uint8)
    ex.combine(part, gathered, result)
    return result.numpy().view(np.uint64).copy()


def point_sharded_msm(local_msm: Callable[[np.ndarray, np.ndarray], np.ndarray], sum_points: Callable[[np.ndarray], np.ndarray],
                      bases: np.ndarray, scalars: np.ndarray, group=None) -> np.ndarray:
    """Point-sharded MSM over host arrays.  Every rank passes the FULL bases/scalars views (or at least
    its own slice filled in); only the rank's slice is touched.  Returns the combined Jacobian point,
    identical on all ranks."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(bases.shape[0], world, rank)
    return _combine_np(local_msm(bases[lo:hi], scalars[lo:hi]), sum_points, group)


def window_split_msm(partial_msm: Callable[[int, int], np.ndarray], sum_points: Callable[[np.ndarray], np.ndarray], group=None) -> np.ndarray:
    """Window-split MSM: `partial_msm(rank, world)` returns this rank's partial sum over the windows
    w = rank (mod world), weights 2^(c*w) included (the engine after reef_msm_ctx_set_window_split(rank,
    world)); the partials are exchanged and added exactly like point-sharded ones."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    return _combine_np(partial_msm(rank, world), sum_points, group)


def sharded_rows(local_rows: Callable[[int, int], np.ndarray], rows: int, width: int, group=None) -> np.ndarray:
    """Row-sharded batch of independent MSMs (HyraxPC::commit, src/backend/commitment.rs:187): rank
    r computes the commitments of rows [lo, hi) with `local_rows(lo, hi)` -> (hi - lo, width) uint64
    (width 12: Jacobian points, 4: compressed commitments); one all-gather returns all `rows`
    commitments, in row order, on every rank."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(rows, world, rank)
    per_rank = -(-rows // world) if rows else 0
    mine = np.zeros((per_rank, width), dtype=np.uint64)
    if hi > lo:
        mine[: hi - lo] = np.ascontiguousarray(local_rows(lo, hi), dtype=np.uint64).reshape(hi - lo, width)
    t = torch.from_numpy(mine.view(np.int64).copy())
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    res = np.zeros((rows, width), dtype=np.uint64)
    for r in range(world):
        a, b = shard_bounds(rows, world, r)
        res[a:b] = out[r].numpy().view(np.uint64)[: b - a]
    return res
