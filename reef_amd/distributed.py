"""One-process-per-GPU sharding of a large MSM.

The path shards by POINTS: rank r owns pairs [r*n/N, (r+1)*n/N) (its slice of the commitment
key stays resident on its GPU, its slice of the scalars is the only per-call upload), computes
a full partial MSM, and the N partial sums -- 96 bytes each -- are exchanged with ONE all-gather
(RCCL over xGMI on the GPUs; RCCL has no elliptic-curve reduction, so the "reduce" is an
all-gather followed by N-1 on-device additions, identical on every rank).  A window split (every
rank keeps all points, owns windows w = r mod N) is provided for comparison: same exchange, but it
replicates the key and the scalar upload, so it is not the default.

Independent units need no split at all (8e.1): the rows of a Hyrax commitment share the row
generators (replicated, resident) and are dealt out in contiguous blocks, `sharded_rows`; the only
exchange is an all-gather of the row commitments.

The arithmetic is injected (`local_msm`, `add_points`): on the GPU box these are the C-ABI calls of
reef_amd.msm; the world_size-2 gloo tests on CPU inject the oracle as a stand-in to check the
sharding, the exchange and the combination order.  There is no CPU fallback in the product path.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced point ranges; the first n % world ranks get one extra pair."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def window_owner(window: int, world: int) -> int:
    return window % world


def all_gather_points(partial: np.ndarray, group=None) -> np.ndarray:
    """All-gather one 96-byte Jacobian point per rank -> (world, 12) uint64 (host tensors; the GPU
    bench gathers device tensors on the MSM's own stream instead, see bench.py)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    t = torch.from_numpy(np.ascontiguousarray(partial, dtype=np.uint64).view(np.int64).reshape(12).copy())
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    return np.stack([o.numpy().view(np.uint64) for o in out])


def sharded_msm(local_msm: Callable[[np.ndarray, np.ndarray], np.ndarray],
                add_points: Callable[[np.ndarray], np.ndarray],
                bases: np.ndarray, scalars: np.ndarray, group=None) -> np.ndarray:
    """Point-sharded MSM.  Every rank passes the FULL bases/scalars views (or at least its own
    slice filled in); only the rank's slice is touched.  Returns the combined Jacobian point,
    identical on all ranks."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(bases.shape[0], world, rank)
    partial = local_msm(bases[lo:hi], scalars[lo:hi])
    gathered = all_gather_points(partial, group)
    return add_points(gathered)           # fixed order 0..N-1 on every rank: bit-identical results


def window_sharded_msm(local_window_sums: Callable[[List[int]], List[np.ndarray]],
                       combine_windows: Callable[[Sequence[np.ndarray]], np.ndarray],
                       n_windows: int, group=None) -> np.ndarray:
    """Window-sharded MSM (north_star's split): rank r computes the window sums S_w for
    w = r mod N; all ranks gather all S_w and run the same Horner combine."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = [w for w in range(n_windows) if window_owner(w, world) == rank]
    sums = local_window_sums(mine)
    per_rank = -(-n_windows // world)
    slots = np.zeros((per_rank, 12), dtype=np.uint64)
    for k, s in enumerate(sums):
        slots[k] = s
    import torch
    t = torch.from_numpy(slots.view(np.int64).copy())
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    all_sums: List[np.ndarray] = [None] * n_windows  # type: ignore
    for r in range(world):
        arr = out[r].numpy().view(np.uint64)
        for k, w in enumerate([w for w in range(n_windows) if window_owner(w, world) == r]):
            all_sums[w] = arr[k]
    return combine_windows(all_sums)


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

