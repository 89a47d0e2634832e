"""GPU read prep (musc_reads_sort_unique) against the reference's contract: bytewise sort of the
prepared reads, collapse of identical sequences (cmd/muscato/main.go sortReads + GNU sort,
cmd/muscato_uniqify/main.go:83-135), restated in oracle.uniqify."""
import os
import random

import numpy as np
import pytest

from oracle import muscato_oracle as orc

from cases import make_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from muscato_amd import Engine
    e = Engine(0)
    yield e
    e.close()


def check_groups(reads, order, ustart):
    """order/ustart describe exactly sorted(set(reads)) with stable groups."""
    uniq = sorted(set(reads))
    assert len(ustart) == len(uniq) + 1 and ustart[0] == 0 and ustart[-1] == len(reads)
    assert sorted(order.tolist()) == list(range(len(reads)))
    for g, u in enumerate(uniq):
        grp = order[ustart[g]:ustart[g + 1]].tolist()
        assert grp, "empty group"
        assert all(reads[i] == u for i in grp)
        assert grp == sorted(grp), "ties must keep input order"
    return uniq


def names_like_reference(reads, names, order, ustart):
    """What the host does with order/ustart: `seq, count, names joined in bytewise name order`."""
    out = []
    for g in range(len(ustart) - 1):
        grp = order[ustart[g]:ustart[g + 1]]
        na = b";".join(sorted(names[i] for i in grp))
        if len(na) > 1000:
            na = na[:996] + b"..."
        out.append((reads[grp[0]], len(grp), na))
    return out


@pytest.mark.parametrize("seed", range(12))
def test_sort_unique_random(eng, seed):
    rng = random.Random(seed)
    alphabet = b"ACGT" if seed % 3 else b"ACGTX"
    maxlen = rng.choice([1, 20, 21, 22, 43, 100, 130])
    pool = [bytes(rng.choice(alphabet) for _ in range(rng.randint(0 if seed == 5 else 1, maxlen))) for _ in range(60)]
    # prefixes of each other, duplicates, near-duplicates differing in the last base
    reads = []
    for _ in range(rng.randint(1, 400)):
        r = rng.choice(pool)
        u = rng.random()
        if u < 0.2 and len(r) > 1:
            r = r[:rng.randint(1, len(r))]
        elif u < 0.3 and r:
            r = r[:-1] + bytes([rng.choice(alphabet)])
        reads.append(r)
    order, ustart = eng.sort_unique_reads(reads)
    uniq = check_groups(reads, order, ustart)
    assert eng.n_reads == len(uniq)
    names = [b">r%d_%d" % (rng.randint(0, 50), i) for i in range(len(reads))]
    got = names_like_reference(reads, names, order, ustart)
    exp = orc.uniqify([r + b"\t" + n for r, n in zip(reads, names)])
    assert got == [(u.seq, u.count, u.names) for u in exp]


def test_sort_unique_edge_cases(eng):
    order, ustart = eng.sort_unique_reads([])
    assert len(order) == 0 and ustart.tolist() == [0] and eng.n_reads == 0
    order, ustart = eng.sort_unique_reads([b"ACGT"])
    assert order.tolist() == [0] and ustart.tolist() == [0, 1]
    order, ustart = eng.sort_unique_reads([b"GATTACA"] * 7)
    assert order.tolist() == list(range(7)) and ustart.tolist() == [0, 7]
    # every key-word boundary (21 bases per word), X last, a prefix before its extensions
    reads = [b"A" * 22, b"A" * 21, b"A" * 21 + b"C", b"A" * 20 + b"X", b"A" * 42 + b"T", b"A" * 42, b"A" * 43,
             b"X", b"T", b"A" * 21 + b"A"]
    order, ustart = eng.sort_unique_reads(reads)
    check_groups(reads, order, ustart)


def test_sort_unique_empty_reads_and_ragged_counts(eng):
    """All-empty reads (maxlen 0: one key word of zeros), and read counts that are not a multiple
    of the 256-thread block on either side of it."""
    order, ustart = eng.sort_unique_reads([b""] * 5)
    assert order.tolist() == [0, 1, 2, 3, 4] and ustart.tolist() == [0, 5] and eng.n_reads == 1
    rng = random.Random(99)
    for n in (255, 256, 257, 70001):
        reads = [bytes(rng.choice(b"ACGT") for _ in range(rng.randint(0, 30))) for _ in range(n)]
        order, ustart = eng.sort_unique_reads(reads)
        check_groups(reads, order, ustart)


@pytest.mark.parametrize("seed", [1, 9, 30])
def test_prepared_reads_give_the_same_hits(eng, seed):
    """Loading through the GPU prep == loading the sorted unique reads directly."""
    from muscato_amd import Config, sorted_hits
    ocfg, ureads, targets = make_case(seed)
    rng = random.Random(seed)
    raw = list(ureads) + [rng.choice(ureads) for _ in range(len(ureads))]
    rng.shuffle(raw)
    cfg = Config(Windows=list(ocfg.Windows), WindowWidth=ocfg.WindowWidth, PMatch=ocfg.PMatch, MinDinuc=ocfg.MinDinuc,
                 MaxReadLength=ocfg.MaxReadLength, MaxMatches=ocfg.MaxMatches, MMTol=ocfg.MMTol, MatchMode=ocfg.MatchMode)
    eng.load_targets(targets)
    eng.load_reads(ureads)
    exp = sorted_hits(eng.match(cfg, apply_mmtol=True))
    order, ustart = eng.sort_unique_reads(raw)
    assert check_groups(raw, order, ustart) == list(ureads)
    got = sorted_hits(eng.match(cfg, apply_mmtol=True))
    assert got.shape == exp.shape and (got == exp).all()


def test_sort_unique_two_million_reads(eng):
    """2 M reads of 100 bp with 10 % duplicates: order and groups against numpy's sort."""
    rng = np.random.default_rng(7)
    n, L = 2_000_000, 100
    R = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L), dtype=np.uint8)]
    dup = rng.integers(0, n, size=n // 10)
    R[rng.integers(0, n, size=n // 10)] = R[dup]
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    buf = np.concatenate([R.reshape(-1), np.zeros(8, np.uint8)])
    order, ustart = eng.sort_unique_reads_arrays(buf.ctypes.data, off.ctypes.data, n, False)
    view = np.ascontiguousarray(R).view([("s", "S%d" % L)]).ravel()
    uniq, first, counts = np.unique(view, return_index=True, return_counts=True)
    assert len(ustart) == len(uniq) + 1
    assert (np.diff(ustart.astype(np.int64)) == counts).all()
    assert (view[order[ustart[:-1]]] == uniq).all()          # group heads are the sorted distinct reads
    assert (order[ustart[:-1]] == first).all()               # and each group's first member is its first occurrence
    assert (view[order] == np.repeat(uniq, counts)).all()     # every member sits in its group
    print("sort+collapse of %d reads: %.1f ms on the device" % (n, eng.stats()["ms_read_prep"]))
