"""Row N4 oracle: the tree of MerkleCommitment::new and the path witnesses, pinned by replaying the reference's own
test `make_mt` (src/backend/merkle_tree.rs:209-257: document [2..8], every leaf's path recomputes the commitment); the
property holds for any node hash, so it pins the TREE, not neptune's digests (stand-in constants, see the oracle's header)."""
from oracle import merkle_oracle as M


def test_reference_make_mt_replayed():
    p = M.standin_params()
    doc = [2, 3, 4, 5, 6, 7, 8]                                   # the reference's "document"
    root, tree = M.commit(doc, p)
    assert [len(l) for l in tree] == [4, 2, 1] and tree[-1][0] == root
    for q in range(len(doc)):                                     # qs = 0..6 in the reference
        wits = M.path_wits(doc, tree, q)
        assert len(wits) == len(tree)
        assert M.root_from_path(doc, q, wits, p) == root
    # the four query shapes of new_parent (merkle_tree.rs:87-103)
    assert tree[0][3] == M.hash_query([6, 8, 0, 0], p)            # odd last leaf: (idx, char, 0, 0)
    assert tree[0][0] == M.hash_query([0, 2, 1, 3], p)
    assert tree[1][1] == M.hash_query([tree[0][2], tree[0][3]], p)
    assert root == M.hash_query([tree[1][0], tree[1][1]], p)


def test_odd_levels_and_single_symbols():
    p = M.standin_params()
    for n in (1, 2, 3, 5, 6, 9, 17, 100):
        doc = [(7 * i + 3) % 131 for i in range(n)]
        root, tree = M.commit(doc, p)
        m, sizes = (n + 1) // 2, []
        while True:
            sizes.append(m)
            if m <= 1:
                break
            m = (m + 1) // 2
        assert [len(l) for l in tree] == sizes
        for q in range(n):
            assert M.root_from_path(doc, q, M.path_wits(doc, tree, q), p) == root
        for h in range(len(tree) - 1):                            # an odd last node is hashed with zero (merkle_tree.rs:97-99)
            if len(tree[h]) % 2:
                assert tree[h + 1][-1] == M.hash_query([tree[h][-1], 0], p)


def test_permutation_is_a_permutation_and_sensitive():
    p = M.standin_params()
    assert len(p.rc) == 5 * 64 and all(0 <= c < M.Q for c in p.rc) and len(set(p.rc)) == len(p.rc)
    a = M.poseidon_permute([1, 2, 3, 4, 5], p)
    b = M.poseidon_permute([1, 2, 3, 4, 6], p)
    assert a != b and len(set(a)) == 5 and all(0 <= x < M.Q for x in a)
    # the MDS matrix is invertible (Cauchy): distinct rows, full rank over the field by construction 1/(x_i + y_j)
    assert all(p.mds[i][j] * (i + 5 + j) % M.Q == 1 for i in range(5) for j in range(5))


def test_oracle_matches_committed_fixture():
    """tests/golden/next_rows_golden.json (oracle/gen_golden_next_rows.py): regression anchors of the oracle on the stand-in
    constants (not neptune's digests)."""
    import json
    import os
    data = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "next_rows_golden.json")))
    for case in data["merkle"]:
        p = M.standin_params(M.Q if case["scalar_field_of"] == "pallas" else M.P, 5, case["rf"], case["rp"])
        root, tree = M.commit(case["doc"], p)
        assert hex(root) == case["root"] and [len(l) for l in tree] == case["level_sizes"] and hex(tree[0][0]) == case["first_leaf"]


def test_product_side_openings_follow_the_reference():
    """reef_amd.merkle.MerkleCommitment mirrors the reference's struct (merkle_tree.rs:11-16) and its openings (path_wits :128-190, make_wits
    :116-126), which are look-ups in the tree on the host.  Here the tree comes from the oracle (no GPU): every path equals the oracle's
    restatement and recomputes the commitment as the reference's own test does (make_mt, :209-257); odd levels, a missing right sibling
    and a single symbol included."""
    import pytest
    from oracle import merkle_oracle as MO
    from reef_amd.merkle import MerkleCommitment, MerkleWit
    from reef_amd.sumcheck import ints_to_array
    p = MO.standin_params()
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31):
        doc = [(i * 7 + 3) % 131 for i in range(n)]
        root, tree = MO.commit(doc, p)
        mc = MerkleCommitment(doc, root, [ints_to_array(lv) for lv in tree])
        assert mc.commitment == root and len(mc.tree) == len(tree)
        for q in range(n):
            w = mc.path_wits(q)
            assert all(isinstance(x, MerkleWit) for x in w) and len(w) == len(tree)
            assert [tuple(x) for x in w] == [tuple(x) for x in MO.path_wits(doc, tree, q)], (n, q)
            assert MO.root_from_path(doc, q, w, p) == mc.commitment
        assert mc.make_wits([0, n - 1]) == [mc.path_wits(0), mc.path_wits(n - 1)]
        with pytest.raises(IndexError):
            mc.path_wits(n)                        # the reference asserts idx < doc.len()
    doc = [2, 3, 4, 5, 6, 7, 8]                    # the reference's own document (merkle_tree.rs:214)
    root, tree = MO.commit(doc, p)
    first = MerkleCommitment(doc, root, [ints_to_array(lv) for lv in tree]).path_wits(6)[0]
    assert first == MerkleWit(True, 0, 0)          # the last symbol of an odd document has no right sibling: (Some(0), 0), :133-139
