"""Full-size properties at BASELINE.json configs[1] (1 M raw 100-bp reads x 100 k targets of
1000 bp, Windows=0,20, WindowWidth=15, PMatch=0.97, MinDinuc=5): things that must hold at any
size and need no oracle.  The oracle-sized cases live in test_gpu_parity.py; bench.py checks a
900 k-read sample of cfg3 against the literal oracle on every run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_TARGETS, TLEN, N_READS, L = 100_000, 1000, 900_000, 100
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


@pytest.fixture(scope="module")
def workload():
    rng = np.random.default_rng(20261003)
    T = BASES[rng.integers(0, 4, size=(N_TARGETS, TLEN), dtype=np.uint8)]
    ncopy = N_TARGETS // 5  # 20 % mutated copies -> multi-map
    src = rng.integers(0, N_TARGETS - ncopy, size=ncopy)
    T[N_TARGETS - ncopy:] = T[src]
    sub = rng.random((ncopy, TLEN)) < 0.02
    T[N_TARGETS - ncopy:][sub] = BASES[rng.integers(0, 4, size=int(sub.sum()))]
    g = rng.integers(0, N_TARGETS, size=N_READS)
    p = rng.integers(0, TLEN - L + 1, size=N_READS)
    p[rng.random(N_READS) < 0.002] = 0
    p[rng.random(N_READS) < 0.002] = TLEN - L
    R = T[g[:, None], p[:, None] + np.arange(L)[None, :]].copy()
    kind = rng.random(N_READS)
    exact = kind < 0.25                      # planted verbatim
    noisy = (kind >= 0.25) & (kind < 0.8)    # 1 % substitutions
    sub = (rng.random(R.shape) < 0.01) & noisy[:, None]
    R[sub] = BASES[rng.integers(0, 4, size=int(sub.sum()))]
    rnd = kind >= 0.8
    R[rnd] = BASES[rng.integers(0, 4, size=(int(rnd.sum()), L))]
    # unique reads, bytewise sorted (the hot path's input contract); keep the plant bookkeeping
    view = np.ascontiguousarray(R).view([("s", "S%d" % L)]).ravel()
    _, first = np.unique(view, return_index=True)
    R, g, p, exact = R[first], g[first], p[first], exact[first]
    return {"T": T, "R": R, "g": g, "p": p, "exact": exact}


@pytest.fixture(scope="module")
def eng(workload):
    from muscato_amd import Engine
    e = Engine(0)
    T = workload["T"]
    toff = np.arange(N_TARGETS + 1, dtype=np.uint64) * np.uint64(TLEN)
    e.load_targets_arrays(np.concatenate([T.reshape(-1), np.zeros(8, np.uint8)]), toff)
    yield e
    e.close()


def cfg(pmatch=0.97, mmtol=0):
    from muscato_amd import Config
    return Config(Windows=[0, 20], WindowWidth=15, PMatch=pmatch, MinDinuc=5, MaxReadLength=100,
                  MaxMatches=1000000, MMTol=mmtol, MatchMode="best")


def load_reads(eng, R):
    off = np.arange(len(R) + 1, dtype=np.uint64) * np.uint64(L)
    eng.load_reads_arrays(np.concatenate([R.reshape(-1), np.zeros(8, np.uint8)]), off)


def key(h):
    return (h[:, 0].astype(np.uint64) << np.uint64(40)) | (h[:, 1].astype(np.uint64) << np.uint64(12)) | \
        h[:, 2].astype(np.uint64)


def test_properties_at_config2_size(eng, workload):
    from muscato_amd import sorted_hits
    T, R = workload["T"], workload["R"]
    load_reads(eng, R)
    allh = sorted_hits(eng.match(cfg(), apply_mmtol=False))
    st = eng.stats()
    assert st["n_reads"] == len(R) and st["n_overflow_blocks"] == 0
    assert st["n_candidates"] >= st["n_pairs"] >= st["n_accepted"] == len(allh) > 500_000

    # (1) every tuple is real: in range, within budget, and nmiss is the true Hamming distance
    assert (allh[:, 2] + L <= TLEN).all() and (allh[:, 3] <= 3).all()
    sub = T[allh[:, 1][:, None], allh[:, 2][:, None] + np.arange(L)[None, :]]
    assert ((sub != R[allh[:, 0]]).sum(axis=1) == allh[:, 3]).all()

    # (2) no duplicates (the union over windows is a set)
    k = key(allh)
    assert len(np.unique(k)) == len(k)

    # (3) completeness on planted reads: a verbatim read must be reported at its source with
    # nmiss 0 -- unless the reference itself cannot see it: both windows must pass MinDinuc, and at
    # target position 0 window 0 is blind for reads longer than 100-15 (the literal-100 rule),
    # window 1 still finds it
    ex = np.nonzero(workload["exact"])[0]
    want = (ex.astype(np.uint64) << np.uint64(40)) | (workload["g"][ex].astype(np.uint64) << np.uint64(12)) | \
        workload["p"][ex].astype(np.uint64)
    found = np.isin(want, k)
    missing = ex[~found]
    from oracle import muscato_oracle as orc
    ocfg = orc.Config(Windows=[0, 20], WindowWidth=15, MinDinuc=5, MaxReadLength=100)
    for m in missing[:200]:
        r = bytes(R[m])
        assert not (orc.window_valid(r, 0, ocfg) and workload["p"][m] != 0) and not orc.window_valid(r, 1, ocfg), \
            "a planted read with a usable window was not found"
    assert len(missing) < 0.01 * len(ex)

    # (4) idempotence
    again = sorted_hits(eng.match(cfg(), apply_mmtol=False))
    assert again.shape == allh.shape and (again == allh).all()

    # (5) best+MMTol keeps, per read, exactly the tuples within MMTol of that read's best
    best = sorted_hits(eng.match(cfg(mmtol=1), apply_mmtol=True))
    assert np.isin(key(best), k).all()
    bmin = np.full(len(R), 99, dtype=np.int64)
    np.minimum.at(bmin, allh[:, 0], allh[:, 3].astype(np.int64))
    keep = allh[:, 3] <= bmin[allh[:, 0]] + 1
    assert keep.sum() == len(best) and (allh[keep] == best).all()

    # (6) a looser budget can only add tuples
    loose = sorted_hits(eng.match(cfg(pmatch=0.93), apply_mmtol=False))
    assert len(loose) >= len(allh) and np.isin(k, key(loose)).all()


def test_shard_invariance(eng, workload):
    """Reads sharded, database replicated: concatenating the shards' tuples (read_idx rebased)
    equals the single pass -- what the N-GPU path relies on."""
    from muscato_amd import sorted_hits
    from muscato_amd.dist import shard_range
    R = workload["R"]
    load_reads(eng, R)
    whole = sorted_hits(eng.match(cfg(), apply_mmtol=True))
    parts = []
    for rank in range(3):
        lo, hi = shard_range(len(R), rank, 3)
        load_reads(eng, R[lo:hi])
        h = eng.match(cfg(), apply_mmtol=True)
        h[:, 0] += lo
        parts.append(h)
    cat = sorted_hits(np.concatenate(parts))
    assert cat.shape == whole.shape and (cat == whole).all()


@pytest.mark.parametrize("pipeline", ["0", "1"])
def test_many_batches_equal_one_batch(workload, eng, monkeypatch, pipeline):
    """Many small batches: the first pass sizes them one by one, repeated passes run without host
    round trips -- on one stream, or (MUSC_PIPELINE=1) over two streams with k_screen of batch b+1
    beside k_confirm of batch b.  All must give the single-batch tuples."""
    from muscato_amd import Engine, sorted_hits
    R, T = workload["R"], workload["T"]
    load_reads(eng, R)
    whole = sorted_hits(eng.match(cfg(mmtol=1), apply_mmtol=True))
    monkeypatch.setenv("MUSC_BATCH_READS", "70001")  # read at musc_init; not a multiple of the tile
    monkeypatch.setenv("MUSC_PIPELINE", pipeline)
    e2 = Engine(0)
    try:
        toff = np.arange(N_TARGETS + 1, dtype=np.uint64) * np.uint64(TLEN)
        e2.load_targets_arrays(np.concatenate([T.reshape(-1), np.zeros(8, np.uint8)]), toff)
        load_reads(e2, R)
        for rep in range(3):  # pass 0 sizes, passes 1-2 are sync-free
            got = sorted_hits(e2.match(cfg(mmtol=1), apply_mmtol=True))
            assert e2.stats()["n_batches"] == -(-len(R) // 70001)
            assert got.shape == whole.shape and (got == whole).all(), "pass %d differs" % rep
    finally:
        e2.close()
