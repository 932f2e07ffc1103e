"""The SHAPE of the automaton Reef builds for a regex -- states, edges, largest skip offset, longest accepting path -- restated for the
regex family the reference's own benchmark scripts use on large documents: anchors, literal characters and the skips `.`, `.*`,
`.{n}`, `.{a,b}` (tests/scripts/dna.sh:6-13, password.sh; README.md:63).  TEST INFRASTRUCTURE ONLY: the inputs of oracle/costs_oracle.py
(= src/backend/costs.rs), from which oracle/gen_replay_shapes.py derives the MSM lengths of the replay.

What is restated, function by function (paths relative to the Reef repository):
  * top-level anchoring        src/frontend/regex/parser.rs:13-43   r -> .*r.* ; ^r -> r.* ; r$ -> .*r ; ^r$ -> r
  * RegexF::extract_skip       src/frontend/regex/mod.rs:318-350    leading skips are merged into ONE (App case: pa.app(pb))
  * SAFA::new / add / add_skip / add_derivatives   src/frontend/safa.rs:86-209:
        a node per distinct regex (derivatives are hash-consed);
        a node whose regex starts with a skip gets ONE skip edge to the remainder, plus -- unless the skip is `.*` (full) or empty
        (nil) -- one edge with the COMPLEMENT skip to the sink, which is created on the spot with one epsilon self-loop (:104-117);
        any other node gets an epsilon self-loop and one edge per alphabet character (:143-156): to the derivative, which for a
        literal is the rest of the regex on its own character and the empty regex (the sink) on every other one; a sink that is first
        reached this way is a node like any other and gets its own epsilon + |alphabet| self-edges;
        the empty suffix (nil) is accepting and has derivative `empty` on every character.
  * SAFA::num_states / num_edges / max_skip_offset   safa.rs:202, :310, :315-331 (OpenSet::max_offset, openset.rs:381-388: the START of an
        open last range -- so `.*` counts 0 and the complement of `.{k}`, [0,k-1] u [k+1,*), counts k + 1)
  * the solution lengths the cost model is fed (`final_paths`)   src/backend/r1cs.rs:253-335 with normal_add_table / incr_depth
        (src/backend/r1cs_helper.rs:88-356, :404-426): without forall nodes, ONE entry: the largest depth + 1 over the accepting nodes,
        depth = edges from the initial node (every node of this family has one predecessor that is not itself);
  * max_offsets as NFA::new passes it on   r1cs.rs:108-110   max(max_skip_offset, 1) + 2 ; max_branches = max_stack = 1 without forks
        (r1cs.rs:112, :121).
Regexes outside the family (classes, alternation, lookahead, repetition of anything but `.`) are refused, not approximated.
"""
from __future__ import annotations

import re as _re
from dataclasses import dataclass
from typing import List, Optional, Tuple

STAR = (0, None)


@dataclass(frozen=True)
class Shape:
    num_states: int
    num_edges: int
    max_skip_offset: int
    max_offsets: int          # what r1cs.rs hands to the cost model
    path_lens: Tuple[int, ...]
    regex: str
    alphabet_size: int


def tokens_of(regex: str) -> Tuple[List[tuple], bool, bool]:
    """-> (tokens, anchored at the start, anchored at the end); a token is ('lit', ch) or ('skip', lo, hi|None)."""
    s = regex
    start = s.startswith("^")
    if start:
        s = s[1:]
    end = s.endswith("$") and not s.endswith("\\$")
    if end:
        s = s[:-1]
    toks: List[tuple] = []
    i = 0
    while i < len(s):
        c = s[i]
        if c == ".":
            j = i + 1
            if j < len(s) and s[j] == "*":
                toks.append(("skip", 0, None))
                i = j + 1
            elif j < len(s) and s[j] == "{":
                m = _re.match(r"\{(\d+)(?:,(\d*))?\}", s[j:])
                if not m:
                    raise ValueError(f"unsupported repetition at {i} in {regex[:40]!r}")
                lo = int(m.group(1))
                hi: Optional[int] = lo if m.group(2) is None else (int(m.group(2)) if m.group(2) else None)
                toks.append(("skip", lo, hi))
                i = j + m.end()
            else:
                toks.append(("skip", 1, 1))
                i = j
        elif c in "[](){}|*+?\\^$":
            raise ValueError(f"outside the restated family (skips and literals): {c!r} at {i} in {regex[:40]!r}")
        else:
            toks.append(("lit", c))
            i += 1
    return toks, start, end


def anchored(toks: List[tuple], start: bool, end: bool) -> List[tuple]:
    """parser.rs:13-43."""
    out = list(toks)
    if not start:
        out.insert(0, ("skip",) + STAR)
    if not end:
        out.append(("skip",) + STAR)
    return out


def _skip_app(a, b):
    """OpenSet::app of two single ranges: offsets add."""
    lo = a[0] + b[0]
    hi = None if a[1] is None or b[1] is None else a[1] + b[1]
    return (lo, hi)


def _is_full(s):
    return s[0] == 0 and s[1] is None


def _is_nil(s):
    return s == (0, 0)


def _max_offset(ranges):
    """OpenSet::max_offset (openset.rs:381-388) of a sorted list of ranges."""
    lo, hi = ranges[-1]
    return lo if hi is None else max(hi, lo)


def _negate(s):
    """OpenRange::negate (openset.rs:101-124) of a closed range."""
    lo, hi = s
    assert hi is not None
    return [(hi + 1, None)] if lo == 0 else [(0, lo - 1), (hi + 1, None)]


def shape(regex: str, alphabet_size: int, alphabet: Optional[str] = None) -> Shape:
    toks, s, e = tokens_of(regex)
    if alphabet is not None:
        for t in toks:
            if t[0] == "lit" and t[1] not in alphabet:
                raise ValueError(f"{t[1]!r} is not in the alphabet")     # Reef panics (framework.rs:990-992)
    toks = anchored(toks, s, e)
    n = len(toks)
    # nodes: suffix positions 0..n (n = nil) that are REACHED, plus the sink; node -> depth
    depth = {0: 0}
    order = [0]
    edges = 0
    sink = False                  # does the sink exist
    sink_edges = 0
    max_skip = 0
    accepting = []

    def nullable(p):
        return all(t[0] == "skip" and t[1] == 0 for t in toks[p:])

    i = 0
    while i < len(order):
        p = order[i]
        i += 1
        if nullable(p):
            accepting.append(p)
        if p < n and toks[p][0] == "skip":                    # extract_skip: the leading skips merged into one
            sk = (toks[p][1], toks[p][2])
            q = p + 1
            while q < n and toks[q][0] == "skip":
                sk = _skip_app(sk, (toks[q][1], toks[q][2]))
                q += 1
            edges += 1                                         # the skip edge to the remainder
            max_skip = max(max_skip, _max_offset([sk]))
            if not _is_full(sk) and not _is_nil(sk):           # the complement skip to the sink (safa.rs:104-117)
                if not sink:
                    sink = True
                    sink_edges = 1                             # created here with one epsilon self-loop
                edges += 1
                max_skip = max(max_skip, _max_offset(_negate(sk)))
            nxt = q
        else:                                                  # add_derivatives: epsilon + one edge per character
            edges += 1 + alphabet_size
            if not sink:                                       # the sink is first reached as a derivative: it gets derivatives of its own
                sink = True
                sink_edges = 1 + alphabet_size
            if p == n:
                continue                                       # nil: every character leads to the sink
            nxt = p + 1
        if nxt not in depth:
            depth[nxt] = depth[p] + 1
            order.append(nxt)
    num_states = len(order) + (1 if sink else 0)
    num_edges = edges + sink_edges
    path = max(depth[p] for p in accepting) + 1 if accepting else 0
    return Shape(num_states=num_states, num_edges=num_edges, max_skip_offset=max_skip, max_offsets=max(max_skip, 1) + 2,
                 path_lens=(path,), regex=regex, alphabet_size=alphabet_size)
