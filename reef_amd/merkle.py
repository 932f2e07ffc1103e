"""Row N4: the Poseidon Merkle commitment of the document (`--merkle`, src/backend/merkle_tree.rs:25-114) through the
C ABI (reef_merkle_commit).  The Poseidon constants and the two domain tags are the caller's (neptune's, on the Rust
side); this module only marshals them."""
from __future__ import annotations

import ctypes
from typing import List, NamedTuple, Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from ._ffi import REEF_HOST, check
from .msm import curve_id
from .sumcheck import array_to_ints, ints_to_array


class PoseidonParams(ctypes.Structure):
    _fields_ = [("width", ctypes.c_uint32), ("full_rounds", ctypes.c_uint32), ("partial_rounds", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("round_constants", ctypes.c_void_p), ("mds", ctypes.c_void_p), ("tag_leaf", ctypes.c_uint64 * 4), ("tag_node", ctypes.c_uint64 * 4)]


def nodes(n: int) -> int:
    return _ffi.load().reef_merkle_nodes(n)


def commit(curve, doc: Sequence[int], width: int, full_rounds: int, partial_rounds: int, round_constants: Sequence[int],
           mds: Sequence[Sequence[int]], tag_leaf: int, tag_node: int) -> Tuple[int, List[List[int]]]:
    """-> (commitment, tree levels) as integers; inputs as integers (canonical)."""
    lib = _ffi.load()
    rc = ints_to_array(list(round_constants))
    m = ints_to_array([x for row in mds for x in row])
    pp = PoseidonParams(width, full_rounds, partial_rounds, 0, rc.ctypes.data, m.ctypes.data,
                        (ctypes.c_uint64 * 4)(*[(tag_leaf >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]),
                        (ctypes.c_uint64 * 4)(*[(tag_node >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]))
    d = np.ascontiguousarray(np.asarray(doc, dtype=np.uint32))
    n = d.shape[0]
    total = nodes(n) if n else 0
    tree = np.zeros((max(total, 1), 4), dtype=np.uint64)
    root = np.zeros((1, 4), dtype=np.uint64)
    check(lib.reef_merkle_commit(curve_id(curve), ctypes.byref(pp), d.ctypes.data, n, REEF_HOST, False, tree.ctypes.data, REEF_HOST, root.ctypes.data))
    flat = array_to_ints(tree[:total])
    levels, m_, off = [], (n + 1) // 2, 0
    while True:
        levels.append(flat[off:off + m_])
        off += m_
        if m_ <= 1:
            break
        m_ = (m_ + 1) // 2
    return array_to_ints(root)[0], levels


def commit_arrays(curve, doc: np.ndarray, width: int, full_rounds: int, partial_rounds: int, round_constants: Sequence[int],
                  mds: Sequence[Sequence[int]], tag_leaf: int, tag_node: int, want_tree: bool = True,
                  devices: Optional[Sequence[int]] = None, info: Optional[dict] = None) -> Tuple[int, List[np.ndarray]]:
    """The same call for documents too large for Python integers: -> (commitment, levels as (m, 4) uint64 arrays of canonical
    limbs, level 0 first; empty when want_tree is False and only the root is copied back).  devices: the tree in blocks over
    several GPUs of this process (reef_merkle_commit_devices; ordinals may repeat); info["blocks"] says how many blocks it took."""
    lib = _ffi.load()
    rc = ints_to_array(list(round_constants))
    m = ints_to_array([x for row in mds for x in row])
    pp = PoseidonParams(width, full_rounds, partial_rounds, 0, rc.ctypes.data, m.ctypes.data,
                        (ctypes.c_uint64 * 4)(*[(tag_leaf >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]),
                        (ctypes.c_uint64 * 4)(*[(tag_node >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]))
    d = np.ascontiguousarray(np.asarray(doc, dtype=np.uint32))
    n = d.shape[0]
    total = nodes(n)
    tree = np.empty((total, 4), dtype=np.uint64) if want_tree else None
    root = np.zeros((1, 4), dtype=np.uint64)
    if devices is None:
        check(lib.reef_merkle_commit(curve_id(curve), ctypes.byref(pp), d.ctypes.data, n, REEF_HOST, False, tree.ctypes.data if want_tree else None, REEF_HOST,
                                     root.ctypes.data))
    else:
        devs = (ctypes.c_int * len(devices))(*devices)
        blocks = ctypes.c_uint32(0)
        check(lib.reef_merkle_commit_devices(curve_id(curve), ctypes.byref(pp), d.ctypes.data, n, False, devs, len(devices), tree.ctypes.data if want_tree else None,
                                             root.ctypes.data, ctypes.byref(blocks)))
        if info is not None:
            info["blocks"] = blocks.value
    levels: List[np.ndarray] = []
    if want_tree:
        m_, off = (n + 1) // 2, 0
        while True:
            levels.append(tree[off:off + m_])
            off += m_
            if m_ <= 1:
                break
            m_ = (m_ + 1) // 2
    return array_to_ints(root)[0], levels


class MerkleWit(NamedTuple):
    """src/backend/merkle_tree.rs:18-22: one step of an opening path."""
    l_or_r: bool                      # True: the node on the path is the LEFT child
    opposite_idx: Optional[int]       # leaf level only: the sibling's document index (0 when the sibling is missing)
    opposite: int                     # the sibling's value (0 when missing)


class MerkleCommitment:
    """Mirror of the reference's `MerkleCommitment<F>` (src/backend/merkle_tree.rs:11-16): `commitment` (the root), `tree` (the levels,
    leaves' parents first) and `doc`, with the openings Reef takes from it -- `path_wits(idx)` (:128-190) and `make_wits(lookups)`
    (:116-126).  `new` builds the tree on the GPU (reef_merkle_commit, or reef_merkle_commit_devices with `devices`); the openings are
    look-ups in that tree on the host, as in the reference.  Levels are kept as (m, 4) uint64 arrays of canonical limbs: a 64 MiB
    document has 2^27 nodes."""

    def __init__(self, doc, commitment: int, levels: Sequence[np.ndarray]):
        self.doc = np.ascontiguousarray(np.asarray(doc, dtype=np.uint32))
        self.commitment = int(commitment)
        self.tree = [np.asarray(lv, dtype=np.uint64).reshape(-1, 4) for lv in levels]

    @classmethod
    def new(cls, curve, doc, width: int, full_rounds: int, partial_rounds: int, round_constants: Sequence[int], mds: Sequence[Sequence[int]],
            tag_leaf: int, tag_node: int, devices: Optional[Sequence[int]] = None) -> "MerkleCommitment":
        """MerkleCommitment::new(&doc, &pc) (merkle_tree.rs:25-80); the Poseidon constants are the caller's (neptune's, on the Rust side)."""
        d = np.ascontiguousarray(np.asarray(doc, dtype=np.uint32))
        if d.shape[0] == 0:
            raise ValueError("empty document")          # the reference indexes next_level[0] of an empty level: a panic
        root, levels = commit_arrays(curve, d, width, full_rounds, partial_rounds, round_constants, mds, tag_leaf, tag_node, want_tree=True, devices=devices)
        return cls(d, root, levels)

    def _node(self, h: int, i: int) -> int:
        row = self.tree[h][i]
        return int(row[0]) | int(row[1]) << 64 | int(row[2]) << 128 | int(row[3]) << 192

    def path_wits(self, idx: int) -> List[MerkleWit]:
        n = self.doc.shape[0]
        if not 0 <= idx < n:
            raise IndexError(f"document index {idx} out of range (the reference asserts idx < doc.len())")
        if idx % 2 == 0:
            wit = MerkleWit(True, 0, 0) if idx + 1 >= n else MerkleWit(True, idx + 1, int(self.doc[idx + 1]))
        else:
            wit = MerkleWit(False, idx - 1, int(self.doc[idx - 1]))
        wits, quo = [wit], idx // 2
        for h in range(len(self.tree) - 1):
            if quo % 2 == 0:
                wits.append(MerkleWit(True, None, 0 if quo + 1 >= self.tree[h].shape[0] else self._node(h, quo + 1)))
            else:
                wits.append(MerkleWit(False, None, self._node(h, quo - 1)))
            quo //= 2
        return wits

    def make_wits(self, m_lookups: Sequence[int]) -> List[List[MerkleWit]]:
        return [self.path_wits(int(q)) for q in m_lookups]
