"""oracle/safa_shape.py (the shape of the automaton SAFA::new builds for skip + literal regexes) against what the reference holds:
the node numbers in the solver traces of its own tests (src/frontend/safa.rs:574-640) and hand counts worked from the construction
rules (safa.rs:86-209, :310-331; src/backend/r1cs.rs:108-110, :253-335)."""
import pytest

from oracle import safa_shape as S


def test_node_counts_behind_the_references_own_traces():
    # test_safa_match_exact (safa.rs:574-590): ^baa$ over "ab" walks nodes 0 -'b'-> 2 -'a'-> 3 -'a'-> 4: node 1 is the sink the first
    # derivative (by 'a') created, five nodes in all
    sh = S.shape("^baa$", 2, "ab")
    assert sh.num_states == 5 and sh.path_lens == (4,)
    assert sh.num_edges == 4 * 3 + 3                                # four regex nodes and the sink: an epsilon loop and one edge per character each
    # test_safa_match_partial (safa.rs:592-610): baa over "ab" walks 0 -*-> 1 -'b'-> 3 -'a'-> 4 -'a'-> 5; 2 is the sink, 6 the empty suffix behind `.*`
    sh = S.shape("baa", 2, "ab")
    assert sh.num_states == 7
    assert sh.num_edges == 1 + 3 * 3 + 1 + 3 + 3                    # .* edge, three literal nodes, the trailing .* edge, nil, sink
    assert sh.max_skip_offset == 0 and sh.max_offsets == 3          # `.*` counts its START (openset.rs:381-388); r1cs.rs:108-110: max(., 1) + 2
    assert sh.path_lens == (6,)                                     # nodes 0,1,3,4,5,6 on the accepting path: depth 5, + 1 (r1cs_helper.rs:347)


def test_hand_counts_of_the_brca_regexes():
    from oracle.gen_replay_shapes import BRCA1_A, BRCA1_B, brca
    k = (1 << 24) - 10000 + 8129
    sh = S.shape(brca(1 << 24, 8129, [BRCA1_A]), 4, "ACGT")
    # nodes: the root, 60 literal states, `.*`, the empty suffix, the sink = 64; edges: the skip and its complement (2), the sink's epsilon loop (1),
    # 60 x (epsilon + 4 characters), the `.*` edge, the empty suffix's 5 = 309; the complement of .{k} is [0,k-1] u [k+1,*): largest offset k + 1
    assert (sh.num_states, sh.num_edges, sh.max_skip_offset, sh.max_offsets, sh.path_lens) == (64, 309, k + 1, k + 3, (63,))
    sh = S.shape(brca(1 << 24, 5784, BRCA1_B), 4, "ACGT")
    assert [len(x) for x in BRCA1_B] == [71, 88, 140]
    assert (sh.num_states, sh.num_edges, sh.path_lens) == (1 + 299 + 3 + 1 + 1, 2 + 1 + 299 * 5 + 3 + 5, (304,))


def test_the_literals_are_the_ones_the_reference_holds():
    """The regexes are the reference's (tests/scripts/dna.sh), and its 1 MB documents hold the literals where the re-basing rule says."""
    import os
    from oracle.gen_replay_shapes import BRCA1_A, BRCA1_B, BRCA2_LENS
    ref = "/root/reference/tests"
    if not os.path.isdir(ref):
        pytest.skip("the reference tree is not on this box")
    sh = open(os.path.join(ref, "scripts", "dna.sh")).read().splitlines()
    assert ("^.{43052424}" + BRCA1_A) in sh[5] and ("^.{43052424}" + BRCA1_A) in sh[7]                 # dna.sh:6 and :8
    assert ("^.{43050079}" + ".*".join(BRCA1_B)) in sh[6] and ("^.{43050079}" + ".*".join(BRCA1_B)) in sh[10]   # :7 and :11
    import re
    lits = re.search(r'--re "\^\.\{32317478\}([ACGT.*]+)"', sh[12]).group(1).split(".*")                # :13
    assert tuple(len(x) for x in lits) == BRCA2_LENS
    doc = open(os.path.join(ref, "docs", "BRCA1_base1m+primary")).read()
    assert len(doc) == 1010000 and doc.find(BRCA1_A) == 1000000 + 8129 and doc.find(BRCA1_B[0]) == 1000000 + 5784
    assert 43052424 - 8129 == 43050079 - 5784                                                        # one base document, two offsets into the gene region
    doc2 = open(os.path.join(ref, "docs", "BRCA2_base1m+primary")).read()
    assert doc2.find(lits[0]) == 1000000 + 1970


def test_outside_the_family_is_refused():
    for rx in ("baa(a|c)$", "[a-b]", "a+", "(?=a)b", r"\d"):
        with pytest.raises(ValueError):
            S.shape(rx, 128)
    with pytest.raises(ValueError):
        S.shape("^.{3}N", 4, "ACGT")                                 # a character outside the alphabet: Reef panics (framework.rs:990-992)
