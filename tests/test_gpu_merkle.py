"""Row N4 on the GPU: the Poseidon Merkle tree of reef_merkle_commit against the oracle, node for node, with the
caller-supplied constants (stand-ins here; neptune's on the Rust side)."""
import numpy as np
import pytest

from oracle import merkle_oracle as M

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 3, 7, 8, 100, 1001, 4096])
def test_tree_matches_oracle(n, gpu_lib):
    from reef_amd import merkle
    p = M.standin_params()
    doc = [(31 * i + 7) % 131 for i in range(n)]
    root, tree = merkle.commit("pallas", doc, p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node)
    eroot, etree = M.commit(doc, p)
    assert tree == etree and root == eroot


def test_reference_make_mt_on_gpu(gpu_lib):
    """merkle_tree.rs:209-257 with the tree built on the GPU: every leaf's path recomputes the commitment."""
    from reef_amd import merkle
    p = M.standin_params()
    doc = [2, 3, 4, 5, 6, 7, 8]
    root, tree = merkle.commit("pallas", doc, p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node)
    for q in range(len(doc)):
        assert M.root_from_path(doc, q, M.path_wits(doc, tree, q), p) == root


def test_other_constants_vesta_field_and_errors(gpu_lib):
    from reef_amd import merkle, msm
    p = M.standin_params(M.P, 5, 8, 20)                           # another field, other round numbers: nothing is hard-wired
    p.tag_leaf, p.tag_node = 12345, M.P - 1
    doc = list(range(50, 77)) + [0xFFFFFFFF]                      # 32-bit symbols
    root, tree = merkle.commit("vesta", doc, p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node)
    assert (root, tree) == M.commit(doc, p)
    with pytest.raises(msm.ReefError):
        merkle.commit("pallas", [1, 2], 3, 8, 56, p.rc, p.mds, 1, 2)      # only width 5 is built
    with pytest.raises(msm.ReefError):
        merkle.commit("pallas", [1, 2], 5, 7, 56, p.rc, p.mds, 1, 2)      # odd number of full rounds


def test_large_document_properties(gpu_lib):
    """2^20 symbols: spot-checked leaves and the whole upper tree against the oracle (the oracle hashes ~2000 nodes here)."""
    from reef_amd import merkle
    p = M.standin_params()
    n = (1 << 20) - 3
    rng = np.random.default_rng(5)
    doc = rng.integers(0, 131, size=n, dtype=np.uint32)
    root, tree = merkle.commit("pallas", doc, p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node)
    for i in (0, 1, 12345, (n + 1) // 2 - 1):
        right = [2 * i + 1, int(doc[2 * i + 1])] if 2 * i + 1 < n else [0, 0]
        assert tree[0][i] == M.hash_query([2 * i, int(doc[2 * i])] + right, p)
    lvl = 10                                                      # from 1024 nodes up, every node
    for h in range(lvl, len(tree) - 1):
        below = tree[h]
        for i in range(len(tree[h + 1])):
            r = below[2 * i + 1] if 2 * i + 1 < len(below) else 0
            assert tree[h + 1][i] == M.hash_query([below[2 * i], r], p)
    assert tree[-1][0] == root
    q = 777
    assert M.root_from_path([int(v) for v in doc[:0]] or doc.tolist(), q, M.path_wits(doc.tolist(), tree, q), p) == root


def test_cfg5_document_size(gpu_lib):
    """BASELINE.json configs[4]: a 64 MiB document is 2^26 + 2 symbols padded to 2^27 (src/backend/framework.rs:997-1008;
    tree: src/backend/merkle_tree.rs:25-80).  The whole tree (2^27 - 1 nodes, 4 GiB) comes back to the host; the oracle hashes
    a sample of it: leaves, interior nodes of every level, and every node from 1024 nodes up."""
    from reef_amd import merkle
    p = M.standin_params()
    n = 1 << 27
    rng = np.random.default_rng(27)
    doc = rng.integers(0, 131, size=n, dtype=np.uint32)
    root, tree = merkle.commit_arrays("pallas", doc, p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node)
    assert [len(l) for l in tree] == [1 << (26 - h) for h in range(27)]

    def val(level, i):
        return sum(int(x) << (64 * k) for k, x in enumerate(tree[level][i]))
    for i in [0, 1, (1 << 26) - 1] + [int(x) for x in rng.integers(0, 1 << 26, size=24)]:
        assert val(0, i) == M.hash_query([2 * i, int(doc[2 * i]), 2 * i + 1, int(doc[2 * i + 1])], p)
    for h in range(0, 16):                                        # interior nodes of the big levels, sampled
        m = len(tree[h + 1])
        for i in [0, m - 1] + [int(x) for x in rng.integers(0, m, size=6)]:
            assert val(h + 1, i) == M.hash_query([val(h, 2 * i), val(h, 2 * i + 1)], p)
    for h in range(16, 26):                                       # 1024 nodes and fewer: every node
        for i in range(len(tree[h + 1])):
            assert val(h + 1, i) == M.hash_query([val(h, 2 * i), val(h, 2 * i + 1)], p)
    assert val(26, 0) == root
    root_only, no_tree = merkle.commit_arrays("pallas", doc, p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node, want_tree=False)
    assert root_only == root and no_tree == []


@pytest.mark.parametrize("rp", [0, 1, 2, 8, 9, 17, 57])
def test_sparse_partial_rounds_equal_the_dense_definition(rp, gpu_lib):
    """The GPU runs the partial rounds in their sparse form (constants derived on the host); the oracle applies the
    permutation as defined.  Every count of partial rounds around the edges of the derivation: none, the dense last
    round alone, one sparse round, the renormalisation period of 8, an odd tail."""
    from reef_amd import merkle
    for field, curve in ((M.Q, "pallas"), (M.P, "vesta")):
        p = M.standin_params(field, 5, 8, rp)
        doc = [(13 * i + 5) % 257 for i in range(21)]
        root, tree = merkle.commit(curve, doc, p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node)
        assert (root, tree) == M.commit(doc, p)


def test_dense_form_kept_behind_the_switch(gpu_lib):
    """REEF_POSEIDON_DENSE=1 (an experiment switch, read once per process: the experiment build in a child) keeps the defining dense rounds: same tree."""
    import os
    import subprocess
    import sys
    from reef_amd import _ffi
    code = ("import sys; sys.path.insert(0, '.');\n"
            "from oracle import merkle_oracle as M\nfrom reef_amd import merkle\n"
            "p = M.standin_params(); doc = list(range(2, 40))\n"
            "assert merkle.commit('pallas', doc, p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node) == M.commit(doc, p)\nprint('dense-ok')\n")
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, REEF_POSEIDON_DENSE="1", REEF_MSM_LIB=_ffi.EXPERIMENT_LIB_PATH), capture_output=True, text=True, timeout=300,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert "dense-ok" in out.stdout, out.stderr[-2000:]


def test_gpu_matches_committed_fixture(gpu_lib):
    import json
    import os
    from reef_amd import merkle
    data = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "next_rows_golden.json")))
    for case in data["merkle"]:
        p = M.standin_params(M.Q if case["scalar_field_of"] == "pallas" else M.P, 5, case["rf"], case["rp"])
        root, tree = merkle.commit(case["scalar_field_of"], case["doc"], p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node)
        assert hex(root) == case["root"] and [len(l) for l in tree] == case["level_sizes"] and hex(tree[0][0]) == case["first_leaf"]


@pytest.mark.parametrize("members", [2, 3, 8])
def test_tree_in_blocks_over_several_devices_is_the_same_tree(members, gpu_lib):
    """reef_merkle_commit_devices (include/reef_msm.h 3d): the bottom level cut into power-of-two blocks, one per member (ordinals
    repeat on a one-GPU box), the levels above hashed from the blocks' roots.  Node for node the oracle's tree for small ragged
    documents (every way the last block can be short: one symbol, one node, an odd node count at every level), node for node
    the one-device tree at 2^20 - 3 symbols; the root alone (no tree copied back); documents smaller than the member count."""
    from reef_amd import merkle, msm
    p = M.standin_params()
    for n in (1, 2, 3, 5, 8, 9, 31, 33, 100, 257, 1001, 4097):
        doc = [(37 * i + 11) % 131 for i in range(n)]
        info = {}
        root, tree = merkle.commit_arrays("pallas", np.asarray(doc, dtype=np.uint32), p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node,
                                          devices=[0] * members, info=info)
        eroot, etree = M.commit(doc, p)
        from reef_amd.sumcheck import array_to_ints
        assert [array_to_ints(lv) for lv in tree] == etree and root == eroot, n
        m0, L = (n + 1) // 2, 0
        while -(-m0 // (1 << L)) > members:
            L += 1
        assert info["blocks"] == -(-m0 // (1 << L))               # the smallest power-of-two block that leaves at most `members` blocks
        if m0 >= members:
            assert info["blocks"] > members // 2                  # ... which leaves fewer than half the members idle
    n = (1 << 20) - 3
    doc = np.random.default_rng(5).integers(0, 131, size=n, dtype=np.uint32)
    root1, tree1 = merkle.commit_arrays("pallas", doc, p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node)
    info = {}
    root, tree = merkle.commit_arrays("pallas", doc, p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node, devices=[0] * members, info=info)
    assert root == root1 and len(tree) == len(tree1) and all((a == b).all() for a, b in zip(tree, tree1))
    assert info["blocks"] == {2: 2, 3: 2, 8: 8}[members]
    root_only, none = merkle.commit_arrays("pallas", doc, p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node, want_tree=False, devices=[0] * members)
    assert root_only == root1 and none == []
    with pytest.raises(msm.ReefError):
        merkle.commit_arrays("pallas", doc[:10], p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node, devices=[0, 99])


@pytest.mark.parametrize("n,devices", [(7, None), (64, None), (1001, None), (1001, [0, 0, 0]), (4097, [0, 0])])
def test_merkle_commitment_mirror_builds_on_the_gpu(n, devices, gpu_lib):
    """MerkleCommitment::new(&doc, &pc) as Reef calls it (merkle_tree.rs:25), then make_wits over lookups (:116): the tree is the GPU's,
    the openings are the product's host look-ups; every path recomputes the commitment with the oracle's sponge, as the reference's
    make_mt test does (:222-249), and the commitment is the oracle's."""
    from reef_amd.merkle import MerkleCommitment
    p = M.standin_params()
    doc = [(31 * i + 7) % 131 for i in range(n)]
    mc = MerkleCommitment.new("pallas", doc, p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node, devices=devices)
    eroot, etree = M.commit(doc, p)
    assert mc.commitment == eroot and [lv.shape[0] for lv in mc.tree] == [len(lv) for lv in etree]
    lookups = sorted({0, 1, n // 2, n - 2 if n > 1 else 0, n - 1})
    for q, wits in zip(lookups, mc.make_wits(lookups)):
        assert [tuple(w) for w in wits] == [tuple(w) for w in M.path_wits(doc, etree, q)]
        assert M.root_from_path(doc, q, wits, p) == mc.commitment
    with pytest.raises(ValueError):
        MerkleCommitment.new("pallas", [], p.t, p.rf, p.rp, p.rc, p.mds, p.tag_leaf, p.tag_node)
