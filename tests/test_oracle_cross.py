"""Cross-check the two independent CPU restatements against each other.

literal  = oracle/literal.cpp  (stage-by-stage: windows, Bloom + rolling hash
           screen, bytewise sorts, block merge-join, searchpairs)
direct   = oracle/muscato_oracle.py::match_direct (set formulation D1-D7)

Covers what the reference fixtures do not pin (SURVEY.md 8c "unpinned"):
nmiss>0, X bases, ragged read lengths, pos-0 hits, the literal-100 rule,
MinDinuc gating, MaxMatches overflow in both modes.  No GPU needed.
"""
import json
import os
import random

import pytest

from oracle import literal
from oracle import muscato_oracle as orc

from cases import make_case, rand_seq


@pytest.mark.parametrize("seed", range(120))
def test_literal_equals_direct(seed):
    cfg, reads, targets = make_case(seed)
    direct = sorted(orc.match_direct(reads, targets, cfg))
    lit = literal.match_literal(reads, targets, cfg, bloom_size=5003 + seed, num_hash=3)
    assert lit == direct
    # a big Bloom filter (no false positives) gives the same answer
    lit2 = literal.match_literal(reads, targets, cfg, bloom_size=4_000_000, num_hash=20, nthreads=3)
    assert lit2 == direct


def test_some_cases_have_mismatches_and_pos0():
    n_mm = n_p0 = n_x = 0
    for seed in range(120):
        cfg, reads, targets = make_case(seed)
        hits = orc.match_direct(reads, targets, cfg)
        n_mm += sum(1 for h in hits if h[3] > 0)
        n_p0 += sum(1 for h in hits if h[2] == 0)
        n_x += sum(1 for h in hits if b"X" in reads[h[0]])
    assert n_mm > 50 and n_p0 > 50 and n_x > 10


def test_literal_100_rule():
    # cmd/muscato_screen/main.go:303-314: at target position 0 the stored right
    # tail ends at absolute index 100-q2, so with ww=15 a read longer than 85 bp
    # cannot be placed at p=0 through window 0 ...
    rng = random.Random(5)
    t = rand_seq(rng, 300, b"ACGT")
    cfg = orc.Config(Windows=[0], WindowWidth=15, PMatch=1.0, MaxReadLength=100)
    r86, r85 = t[:86], t[:85]
    for reads, expect in (([r86], []), ([r85], [(0, 0, 0, 0)])):
        assert sorted(orc.match_direct(reads, [t], cfg)) == expect
        assert literal.match_literal(reads, [t], cfg) == expect
    # ... but is found at p=0 through a later window (general path)
    cfg2 = orc.Config(Windows=[0, 20], WindowWidth=15, PMatch=1.0, MaxReadLength=100)
    assert sorted(orc.match_direct([t[:100]], [t], cfg2)) == [(0, 0, 0, 0)]
    assert literal.match_literal([t[:100]], [t], cfg2) == [(0, 0, 0, 0)]
    # and one base further in, window 0 works for a 100 bp read
    assert literal.match_literal([t[1:101]], [t], cfg) == [(0, 0, 1, 0)]


def test_x_matches_x_and_counts_against_bases():
    # cmd/muscato_confirm/main.go:151-159: byte inequality, so X==X
    t = b"ACGTACGTXXACGTTTGACA"
    cfg = orc.Config(Windows=[0], WindowWidth=4, PMatch=0.8, MaxReadLength=50)
    reads = [b"ACGTACGTXXAC", b"ACGTACGTAXAC", b"ACGTXCGTXXAC"]
    exp = [(0, 0, 0, 0), (1, 0, 0, 1), (2, 0, 0, 1)]
    assert sorted(orc.match_direct(reads, [t], cfg)) == exp
    assert literal.match_literal(reads, [t], cfg) == exp


def test_target_end_rejection_no_clipping():
    # cmd/muscato_confirm/main.go:201-203
    t = b"GGGGGGACGTACGTAA"
    # read 0 overhangs the target end by one base at p=10 (first 6 bases match):
    # rejected, never clipped; read 1 ends flush with the target end: accepted.
    cfg = orc.Config(Windows=[0], WindowWidth=4, PMatch=1.0, MaxReadLength=50)
    reads = [b"ACGTAAT", b"CGTAA"]
    exp = [(1, 0, 11, 0)]
    assert sorted(orc.match_direct(reads, [t], cfg)) == exp
    assert literal.match_literal(reads, [t], cfg) == exp


@pytest.mark.parametrize("mode", ["first", "best"])
def test_maxmatches_overflow_literal(mode):
    # cmd/muscato_confirm/main.go:233-242 ("first" keeps MaxMatches+1) and
    # :424-448 ("best": heap + tail truncation).  One key block, 12 targets.
    key = b"ACGT"
    targets = [key + bytes([b"ACGT"[(i >> s) & 3] for s in (0, 2)]) + b"AAAA" for i in range(12)]
    reads = [key + b"AA"]
    cfg = orc.Config(Windows=[0], WindowWidth=4, PMatch=0.5, MaxReadLength=50, MaxMatches=5, MatchMode=mode)
    with pytest.raises(OverflowError):
        orc.match_direct(reads, targets, cfg)
    lit = literal.match_literal(reads, targets, cfg)
    full = sorted(orc.match_direct(reads, targets, cfg, check_overflow=False))
    assert set(lit) <= set(full)
    assert len(lit) == (6 if mode == "first" else 5)
    if mode == "first":
        # candidates arrive in bytewise order of "key\tleft\tright\tgene\tpos"
        order = sorted(range(12), key=lambda g: (targets[g][4:], g))
        assert sorted(h[1] for h in lit) == sorted(order[:6])


MUSCATO_CASES = [("00", False), ("01", False), ("02", False), ("03", False), ("04", True)]


@pytest.mark.parametrize("case,rev", MUSCATO_CASES)
def test_literal_reproduces_fixture(golden_dir, case, rev):
    d = os.path.join(golden_dir, "muscato", case)
    with open(os.path.join(d, "config.json"), "rb") as f:
        cfg = orc.Config.from_json(json.loads(f.read()))
    seqs, ids = orc.prep_targets_file(os.path.join(d, "genes.txt"), rev)
    with open(os.path.join(d, "reads.fastq"), "rb") as f:
        ureads = orc.uniqify(orc.prep_reads(orc.read_fastq(f.read()), cfg))
    hits = literal.match_literal([u.seq for u in ureads], seqs, cfg, bloom_size=4000000, num_hash=20)
    res = orc.results_text(hits, ureads, seqs, ids, cfg)
    with open(os.path.join(d, "result_e.txt"), "rb") as f:
        assert res == f.read()
