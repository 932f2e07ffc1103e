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


# ---- independent units over devices (round 4) -----------------------------------------------------------------
# Reef's own MSMs are 2^14..2^16 points and must not be split: what a multi-GPU node can take from a --prove run is
# WHOLE units that do not depend on each other -- the final SNARK's three arguments (the two Spartan inner-product arguments and
# the consistency argument, src/backend/framework.rs:695-721), the rows of the document commitment (`sharded_rows`) -- and, for
# memory rather than time, the two curves of a folding step (their commitments depend on each other through the step circuits,
# include/reef_msm.h reef_msm_multi: placing the secondary curve's key on another device frees HBM, it does not overlap them).
def place_units(costs: Sequence[float], world: int) -> List[int]:
    """owner[u] for independent units of the given relative costs: longest unit first onto the least loaded rank (ties: lowest
    rank), deterministic, identical on every rank.  Three arguments of costs (5.4, 4.4, 2.9) ms on 2 ranks -> [0, 1, 1]."""
    load = [0.0] * max(world, 1)
    owner = [0] * len(costs)
    for u in sorted(range(len(costs)), key=lambda k: (-costs[k], k)):
        r = min(range(len(load)), key=lambda k: (load[k], k))
        owner[u] = r
        load[r] += costs[u]
    return owner


def run_placed_units(run_unit: Callable[[int], np.ndarray], widths: Sequence[int], costs: Sequence[float], group=None) -> List[np.ndarray]:
    """Every rank runs the units `place_units` gives it (`run_unit(u)` -> uint64 array of widths[u] words: an argument's L/R
    points, a commitment ...) and ONE all-gather hands every result to every rank, in unit order.  Payloads are a few KiB."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    owner = place_units(costs, world)
    per_rank = [sum(widths[u] for u in range(len(widths)) if owner[u] == r) for r in range(world)]
    buf = np.zeros(max(max(per_rank), 1), dtype=np.uint64)
    off = 0
    for u in range(len(widths)):
        if owner[u] == rank:
            res = np.ascontiguousarray(run_unit(u), dtype=np.uint64).reshape(-1)
            if res.shape[0] != widths[u]:
                raise ValueError(f"unit {u} returned {res.shape[0]} words, {widths[u]} expected")
            buf[off:off + widths[u]] = res
            off += widths[u]
    t = torch.from_numpy(buf.view(np.int64).copy())
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t, group=group)
    offs = [0] * world
    results: List[np.ndarray] = []
    for u in range(len(widths)):
        r = owner[u]
        results.append(out[r].numpy().view(np.uint64)[offs[r]:offs[r] + widths[u]].copy())
        offs[r] += widths[u]
    return results


class LowBitShardedSumCheck:
    """The nlookup sum-check of one folding step (row N2; src/backend/r1cs_helper.rs:441-544) with the table sharded over the ranks
    by the LOW index bits: rank r keeps the entries i with i mod N = r (N a power of two, k = log2 N), as a local table of 2^(ell-k)
    entries indexed by i >> k.  A round folds the TOP index bit (pairs b, b + pow), which both live on the same rank, so every fold is
    rank-local; a round exchanges the three partial sums (3 x 32 B per rank, one all-gather), every rank adds them and derives the SAME
    challenge from the same transcript.  After ell - k rounds a rank holds one entry of T and of EQ; the last k rounds run on the N
    gathered entries, identically on every rank.

    gen_eq_table for the shard: EQ[i] = masses + rs_last * prod_j eq(bit_j(i), last_q[j]) (:508-544) with bit j paired with
    last_q[j]; the k low bits of a local entry are the rank's, so their factors are a constant of the rank that goes into rs_last,
    and local bit j' pairs with last_q[j' + k].  A mass at q belongs to rank q mod N, at local index q >> k.

    `engine`: an object with the methods of reef_amd.sumcheck.SumCheck for a table of 2^(ell-k) entries (on the GPU: that class;
    in the CPU tests: the oracle behind the same names).  `finish_rounds(t, e, rounds) -> transcript tail` runs the last k rounds
    on N-entry tables (the oracle's linear_mle functions on the host: 2 N entries)."""

    def __init__(self, engine, ell: int, modulus: int, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.k = self.world.bit_length() - 1
        if 1 << self.k != self.world or self.k > ell:
            raise ValueError("the low-bit shard needs a power-of-two number of ranks, at most 2^ell")
        self.engine, self.ell, self.q = engine, ell, modulus

    def local_table(self, table: Sequence[int]) -> List[int]:
        """The rank's shard of a full table (zero-padded to 2^ell as the reference pads, r1cs.rs:2323-2330)."""
        n = 1 << self.ell
        return [table[i] if i < len(table) else 0 for i in range(self.rank, n, self.world)]

    def set_table(self, table: Sequence[int]) -> None:
        self.engine.set_table(0, self.local_table(table))

    def gen_eq_table(self, rs: Sequence[int], qs: Sequence[int], last_q: Sequence[int]) -> None:
        q, k = self.q, self.k
        const = rs[len(qs)] % q
        for j in range(k):
            bit = (self.rank >> j) & 1
            const = const * (last_q[j] if bit else (1 - last_q[j])) % q
        mine = [(qq >> k, rr) for qq, rr in zip(qs, rs) if qq & (self.world - 1) == self.rank]
        self.engine.gen_eq_table([rr for _, rr in mine] + [const], [qq for qq, _ in mine], list(last_q[k:]))

    def _allsum3(self, triple: Sequence[int]) -> Tuple[int, int, int]:
        import torch
        words = np.zeros(12, dtype=np.uint64)
        for a, v in enumerate(triple):
            for j in range(4):
                words[4 * a + j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
        t = torch.from_numpy(words.view(np.int64).copy())
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        tot = [0, 0, 0]
        for o in out:
            w = o.numpy().view(np.uint64)
            for a in range(3):
                tot[a] += sum(int(w[4 * a + j]) << (64 * j) for j in range(4))
        return tot[0] % self.q, tot[1] % self.q, tot[2] % self.q

    def run_step(self, challenge: Callable[[int, int, int, int], int]) -> Tuple[List[Tuple[int, int, int]], List[int], int, int]:
        """All ell rounds.  `challenge(i, xsq, x, con)` is the host's transcript (Poseidon sponge; every rank holds the same one).
        -> (coefficient triples per round, challenges, T~(r), EQ~(r)), identical on every rank."""
        import torch
        ell, k, eng = self.ell, self.k, self.engine
        local = ell - k
        coeffs, rs_out = [], []
        g = eng.round_coeffs(1) if local >= 1 else None
        for i in range(1, local + 1):
            tot = self._allsum3(g)
            r = challenge(i, *tot)
            coeffs.append(tot)
            rs_out.append(r)
            if i < local:
                g = eng.fold_and_next_coeffs(i, r)
            else:
                eng.fold(i, r)
        t_loc = eng.read(0, 1)[0] if local >= 1 else eng.read(0, 1)[0]
        e_loc = eng.read(1, 1)[0]
        # the last k rounds: the N surviving entries, entry r on rank r (index bits = the rank), gathered and finished everywhere
        words = np.zeros(8, dtype=np.uint64)
        for a, v in enumerate((t_loc, e_loc)):
            for j in range(4):
                words[4 * a + j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
        t = torch.from_numpy(words.view(np.int64).copy())
        out = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t, group=self.group)
        tt, ee = [], []
        for o in out:
            w = o.numpy().view(np.uint64)
            tt.append(sum(int(w[j]) << (64 * j) for j in range(4)))
            ee.append(sum(int(w[4 + j]) << (64 * j) for j in range(4)))
        for i in range(1, k + 1):
            pow_ = 1 << (k - i)
            xsq = x = con = 0
            for b in range(pow_):
                ts, es = tt[b + pow_] - tt[b], ee[b + pow_] - ee[b]
                xsq += ts * es
                x += es * tt[b] + ts * ee[b]
                con += tt[b] * ee[b]
            tot = (xsq % self.q, x % self.q, con % self.q)
            r = challenge(local + i, *tot)
            coeffs.append(tot)
            rs_out.append(r)
            for b in range(pow_):
                tt[b] = (tt[b] * (1 - r) + tt[b + pow_] * r) % self.q
                ee[b] = (ee[b] * (1 - r) + ee[b + pow_] * r) % self.q
        return coeffs, rs_out, tt[0], ee[0]
