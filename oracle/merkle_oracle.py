"""CPU restatement of Reef's Poseidon Merkle commitment (row N4) -- TEST INFRASTRUCTURE, never imported by the product.

Follows src/backend/merkle_tree.rs of eniac/Reef:
    MerkleCommitment::new   :25-80    leaves (idx, char) pairs, parents of pairs, odd tails padded with zero
    new_parent              :82-114   the four query shapes
    path_wits               :128-190  the sibling witnesses of a leaf
and the shape of the reference's own test `make_mt` (:209-257): recomputing the path of every leaf from its witnesses
gives the commitment.  That test pins the TREE (which nodes hash what) for any hash function; it is replayed with the
reference's inputs in tests/test_merkle_oracle.py.

The hash itself is neptune's Poseidon sponge (Sponge<F, U4>, Mode::Simplex, IOPattern [Absorb(k), Squeeze(1)]), a crate
that is NOT in the reference tree [R]: one absorb-then-squeeze is one permutation of the width-5 state
[tag, x_1..x_k, 0..] whose element 1 is the digest.  `poseidon_permute` is the published permutation (x^5 S-box, R_F/2
full rounds, R_P partial rounds, R_F/2 full rounds; round = add constants, S-box, MDS).  The constants below are
STAND-INS produced by a Grain-LFSR generator written from the published procedure as recalled; the product takes
neptune's constants and tags from its caller (include/reef_msm.h, reef_poseidon_params), so no test here claims to
reproduce neptune's digests: HASH PARITY UNPINNED, tree parity pinned.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

Q = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001   # Pallas scalar field (r1cs_helper.rs:37-38)
P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001


class Params:
    def __init__(self, modulus: int, width: int, rf: int, rp: int, rc: Sequence[int], mds: Sequence[Sequence[int]], tag_leaf: int, tag_node: int):
        self.m, self.t, self.rf, self.rp = modulus, width, rf, rp
        self.rc, self.mds, self.tag_leaf, self.tag_node = list(rc), [list(r) for r in mds], tag_leaf, tag_node


def _grain_bits(n_bits: int, t: int, rf: int, rp: int):
    state = []
    def put(v, w): state.extend((v >> (w - 1 - i)) & 1 for i in range(w))
    put(1, 2); put(0, 4); put(n_bits, 12); put(t, 12); put(rf, 10); put(rp, 10); put((1 << 30) - 1, 30)
    def step():
        b = state[62] ^ state[51] ^ state[38] ^ state[23] ^ state[13] ^ state[0]
        state.pop(0); state.append(b)
        return b
    for _ in range(160):
        step()
    while True:
        if step():
            yield step()
        else:
            step()


def standin_params(modulus: int = Q, width: int = 5, rf: int = 8, rp: int = 56) -> Params:
    """Stand-in Poseidon parameters (NOT neptune's): Grain-LFSR round constants, Cauchy MDS 1/(x_i + y_j), fixed tags."""
    bits = _grain_bits(modulus.bit_length(), width, rf, rp)
    rc = []
    while len(rc) < width * (rf + rp):
        v = 0
        for _ in range(modulus.bit_length()):
            v = (v << 1) | next(bits)
        if v < modulus:
            rc.append(v)
    mds = [[pow(i + width + j, -1, modulus) for j in range(width)] for i in range(width)]
    return Params(modulus, width, rf, rp, rc, mds, tag_leaf=(1 << 64) + 4, tag_node=(1 << 64) + 2)


def poseidon_permute(state: List[int], p: Params) -> List[int]:
    s, m, t = list(state), p.m, p.t
    half = p.rf // 2
    for r in range(p.rf + p.rp):
        s = [(x + p.rc[r * t + i]) % m for i, x in enumerate(s)]
        if r < half or r >= half + p.rp:
            s = [pow(x, 5, m) for x in s]
        else:
            s[0] = pow(s[0], 5, m)
        s = [sum(s[i] * p.mds[i][j] for i in range(t)) % m for j in range(t)]
    return s


def hash_query(query: Sequence[int], p: Params) -> int:
    """One absorb(len(query)) + squeeze(1) of the sponge: merkle_tree.rs:107-113."""
    assert len(query) in (2, 4)
    tag = p.tag_leaf if len(query) == 4 else p.tag_node
    st = [tag] + [x % p.m for x in query] + [0] * (p.t - 1 - len(query))
    return poseidon_permute(st, p)[1]


def new_parent(left: Tuple[Optional[int], int], right: Optional[Tuple[Optional[int], int]], p: Params) -> int:
    """merkle_tree.rs:82-114."""
    (li, lc) = left
    if li is not None and right is not None and right[0] is not None:
        q = [li, lc, right[0], right[1]]
    elif li is not None and right is None:
        q = [li, lc, 0, 0]
    elif li is None and right is not None and right[0] is None:
        q = [lc, right[1]]
    elif li is None and right is None:
        q = [lc, 0]
    else:
        raise ValueError("not a correctly formatted leaf or parent")
    return hash_query(q, p)


def commit(doc: Sequence[int], p: Params) -> Tuple[int, List[List[int]]]:
    """MerkleCommitment::new (merkle_tree.rs:25-80) -> (commitment, tree levels)."""
    tree, level, i = [], [], 0
    while i < len(doc):
        right = (i + 1, doc[i + 1]) if i + 1 < len(doc) else None
        level.append(new_parent((i, doc[i]), right, p))
        i += 2
    tree.append(level)
    while len(level) > 1:
        prev, level, i = level, [], 0
        while i < len(prev):
            right = (None, prev[i + 1]) if i + 1 < len(prev) else None
            level.append(new_parent((None, prev[i]), right, p))
            i += 2
        tree.append(level)
    return level[0], tree


def path_wits(doc: Sequence[int], tree: List[List[int]], idx: int):
    """merkle_tree.rs:128-190 -> list of (l_or_r, opposite_idx or None, opposite)."""
    assert idx < len(doc)
    if idx % 2 == 0:
        w = (True, 0, 0) if idx + 1 >= len(doc) else (True, idx + 1, doc[idx + 1])
    else:
        w = (False, idx - 1, doc[idx - 1])
    wits, quo = [w], idx // 2
    for h in range(len(tree) - 1):
        if quo % 2 == 0:
            wits.append((True, None, 0 if quo + 1 >= len(tree[h]) else tree[h][quo + 1]))
        else:
            wits.append((False, None, tree[h][quo - 1]))
        quo //= 2
    return wits


def root_from_path(doc: Sequence[int], idx: int, wits, p: Params) -> int:
    """The recomputation of the reference's test make_mt (merkle_tree.rs:222-249)."""
    l_or_r, w0, w1 = wits[0]
    q = [idx, doc[idx], w0, w1] if l_or_r else [w0, w1, idx, doc[idx]]
    h = hash_query(q, p)
    for (l_or_r, _, w) in wits[1:]:
        h = hash_query([h, w] if l_or_r else [w, h], p)
    return h
