"""GPU parity tests: the HIP hot path (through the C ABI) against the CPU oracle.

Bit-exact bar: the set of (read, gene, pos, nmiss) tuples must be identical.
Run on the GPU box with ``pytest -m gpu``; everything here goes through
libmuscato_hip.so -- there is no CPU fallback to fall into.
"""
import json
import os
import random

import numpy as np
import pytest

from oracle import literal
from oracle import muscato_oracle as orc

from cases import make_case, mutate, rand_seq

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["auto", "dma", "classic", "lines"])
def eng(request):
    """Every test of this module runs four times: on what the library picks by itself (context
    buckets wherever the run fits them, matched by k_match_t -- comparisons in the lane that owns the
    read, kernels_match_lane.hpp), with MUSC_MATCH=dma (k_match_g, kernels_match_dma.hpp: the second
    fused implementation -- everything from memory by LDS-DMA, three to four waves per SIMD -- where the
    run has two windows, 120-base buckets, records of eight words and no X; k_match_t elsewhere), with
    MUSC_INDEX=classic (the two-kernel path k_screen -> k_confirm on the bucket layout the database's
    density selects: 64-byte buckets for these small databases) and with MUSC_INDEX=lines (the
    two-kernel path on line buckets, the layout of dense databases such as BASELINE config 5)."""
    from muscato_amd import Engine
    old = {k: os.environ.get(k) for k in ("MUSC_INDEX", "MUSC_MATCH")}
    os.environ.pop("MUSC_INDEX", None)
    os.environ.pop("MUSC_MATCH", None)
    if request.param in ("classic", "lines"):
        os.environ["MUSC_INDEX"] = request.param
    elif request.param == "dma":
        os.environ["MUSC_MATCH"] = "dma"
    e = Engine(0)
    e.index_mode = request.param
    yield e
    e.close()
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def to_cfg(ocfg):
    from muscato_amd import Config
    return Config(Windows=list(ocfg.Windows), WindowWidth=ocfg.WindowWidth, PMatch=ocfg.PMatch,
                  MinDinuc=ocfg.MinDinuc, MaxReadLength=ocfg.MaxReadLength, MaxMatches=ocfg.MaxMatches,
                  MMTol=ocfg.MMTol, MatchMode=ocfg.MatchMode)


def as_arr(hits):
    return np.array(sorted(hits), dtype=np.uint32).reshape(-1, 4)


def gpu_hits(eng, ocfg, reads, targets, apply_mmtol):
    from muscato_amd import sorted_hits
    eng.load_targets(targets)
    eng.load_reads(reads)
    return sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=apply_mmtol))


def assert_same(got, exp):
    assert got.shape == exp.shape, "hit count differs: gpu %d vs oracle %d" % (len(got), len(exp))
    assert (got == exp).all()


@pytest.mark.parametrize("seed", range(120))
def test_random_cases_match_oracle(eng, seed):
    """Same seeded cases as tests/test_oracle_cross.py (X bases, ragged lengths, pos-0,
    target-end, 1-3 windows, PMatch 0.7-1, MinDinuc 0-5)."""
    ocfg, reads, targets = make_case(seed)
    full = orc.match_direct(reads, targets, ocfg)
    assert_same(gpu_hits(eng, ocfg, reads, targets, False), as_arr(full))
    from muscato_amd import sorted_hits
    got = sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=True))
    assert_same(got, as_arr(orc.best_filter(full, ocfg.MMTol)))


MUSCATO_CASES = [("00", False), ("01", False), ("02", False), ("03", False), ("04", True)]


@pytest.mark.parametrize("case,rev", MUSCATO_CASES)
def test_reference_fixture_results(eng, golden_dir, case, rev):
    """The reference's own end-to-end fixtures (tests/tests.toml:70-139): hot path on the GPU,
    text prep/post around it, byte-identical results.txt and nonmatch fastq."""
    d = os.path.join(golden_dir, "muscato", case)
    with open(os.path.join(d, "config.json"), "rb") as f:
        ocfg = orc.Config.from_json(json.loads(f.read()))
    seqs, ids = orc.prep_targets_file(os.path.join(d, "genes.txt"), rev)
    with open(os.path.join(d, "reads.fastq"), "rb") as f:
        ureads = orc.uniqify(orc.prep_reads(orc.read_fastq(f.read()), ocfg))
    reads = [u.seq for u in ureads]
    got = gpu_hits(eng, ocfg, reads, seqs, True)
    hits = [tuple(int(x) for x in row) for row in got]
    res = orc.results_text(hits, ureads, seqs, ids, ocfg)
    with open(os.path.join(d, "result_e.txt"), "rb") as f:
        assert res == f.read()
    with open(os.path.join(d, "result.nonmatch_e.txt"), "rb") as f:
        assert orc.nonmatch_text(res, ureads) == f.read()


def test_packed_loaders_equal_ascii_loaders(eng):
    from muscato_amd import sorted_hits
    for seed in (0, 3, 7, 9):  # seeds 0, 3, 9 use the X alphabet
        ocfg, reads, targets = make_case(seed)
        exp = gpu_hits(eng, ocfg, reads, targets, False)
        eng.load_targets_packed(targets)
        eng.load_reads_packed(reads)
        got = sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=False))
        assert_same(got, exp)


def test_packed32_loader_lengths_fixed_and_streamed(eng, monkeypatch):
    """musc_reads_load_packed32 (SURVEY.md 8b's `uint32 lengths_or_offsets` form): u32 lengths with
    and without an X mask against the ASCII loader; reads of one length as the bare 2-bit stream,
    loaded at once and as an asynchronous upload that the match overlaps batch by batch (many small
    batches: every piece boundary is crossed) -- the same tuples every time, and again on the
    repeat (sized) pass over the streamed reads."""
    from muscato_amd import sorted_hits
    from muscato_amd.api import concat, pack_2bit
    for seed in (3, 7, 9, 10):  # seeds 3, 9 use the X alphabet
        ocfg, reads, targets = make_case(seed)
        exp = gpu_hits(eng, ocfg, reads, targets, False)
        buf, off = concat(reads)
        packed, mask = pack_2bit(buf, int(off[-1]))
        packed = np.concatenate([packed, np.zeros(8, np.uint8)])
        lens = np.diff(off.astype(np.int64)).astype(np.uint32)
        eng.load_reads_packed32_ptr(packed.ctypes.data, mask.ctypes.data if mask is not None else 0, lens.ctypes.data, 0, len(reads))
        assert_same(sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=False)), exp)
    # reads of one length, no X
    rng = random.Random(77)
    targets = [rand_seq(rng, 700, b"ACGT") for _ in range(60)]
    rs = set()
    while len(rs) < 3000:
        t = rng.choice(targets)
        p = rng.randint(0, len(t) - 90)
        rs.add(mutate(rng, t[p:p + 90], 0.02, b"ACGT"))
    reads = sorted(rs)
    ocfg = orc.Config(Windows=[0, 22], WindowWidth=11, PMatch=0.95, MinDinuc=2, MaxReadLength=90, MaxMatches=100000, MMTol=1)
    exp = gpu_hits(eng, ocfg, reads, targets, False)
    buf, off = concat(reads)
    packed, mask = pack_2bit(buf, int(off[-1]))
    assert mask is None
    packed = np.concatenate([packed, np.zeros(8, np.uint8)])
    from muscato_amd import Engine
    for batch in ("16777216", "320", "257"):
        monkeypatch.setenv("MUSC_BATCH_READS", batch)  # read at musc_init
        with Engine(0) as e2:
            e2.load_targets(targets)
            for asyn in (False, True):
                e2.load_reads_packed32_ptr(packed.ctypes.data, 0, 0, 90, len(reads), async_upload=asyn)
                for rep in range(2):
                    assert_same(sorted_hits(e2.match(to_cfg(ocfg), apply_mmtol=False)), exp)
            # an upload nobody matches is waited for when the reads are replaced
            e2.load_reads_packed32_ptr(packed.ctypes.data, 0, 0, 90, len(reads), async_upload=True)
            e2.load_reads(reads[:100])
            assert len(e2.match(to_cfg(ocfg), apply_mmtol=False)) > 0


def test_literal_100_rule_on_gpu(eng):
    rng = random.Random(5)
    t = rand_seq(rng, 300, b"ACGT")
    c1 = orc.Config(Windows=[0], WindowWidth=15, PMatch=1.0, MaxReadLength=100)
    c2 = orc.Config(Windows=[0, 20], WindowWidth=15, PMatch=1.0, MaxReadLength=100)
    assert len(gpu_hits(eng, c1, [t[:86]], [t], False)) == 0
    assert_same(gpu_hits(eng, c1, [t[:85]], [t], False), as_arr([(0, 0, 0, 0)]))
    assert_same(gpu_hits(eng, c2, [t[:100]], [t], False), as_arr([(0, 0, 0, 0)]))
    assert_same(gpu_hits(eng, c1, [t[1:101]], [t], False), as_arr([(0, 0, 1, 0)]))


def test_x_semantics_on_gpu(eng):
    t = b"ACGTACGTXXACGTTTGACA"
    c = orc.Config(Windows=[0], WindowWidth=4, PMatch=0.8, MaxReadLength=50)
    reads = [b"ACGTACGTXXAC", b"ACGTACGTAXAC", b"ACGTXCGTXXAC"]
    assert_same(gpu_hits(eng, c, reads, [t], False), as_arr([(0, 0, 0, 0), (1, 0, 0, 1), (2, 0, 0, 1)]))
    # a window made of X only matches X at the same places
    c = orc.Config(Windows=[0], WindowWidth=4, PMatch=1.0, MaxReadLength=50)
    reads = [b"AXXAGG", b"AAAAGG"]
    targets = [b"TTAXXAGGTT", b"TTAAAAGGTT"]
    assert_same(gpu_hits(eng, c, reads, targets, False), as_arr(orc.match_direct(reads, targets, c)))
    assert_same(gpu_hits(eng, c, reads, targets, False), as_arr([(0, 0, 2, 0), (1, 1, 2, 0)]))


def test_empty_and_degenerate_inputs(eng):
    c = orc.Config(Windows=[0, 5], WindowWidth=4, PMatch=1.0, MaxReadLength=50)
    targets = [b"ACGTACGTACGTTTGACA", b"", b"ACG", b"GGGGGGGGGGGG"]
    # no reads at all
    assert len(gpu_hits(eng, c, [], targets, False)) == 0
    # reads shorter than every window; a read that matches nothing
    assert len(gpu_hits(eng, c, [b"ACG", b"A"], targets, True)) == 0
    assert len(gpu_hits(eng, c, [b"TTTTTTTTTT"], targets, True)) == 0
    # empty and too-short targets in the middle keep gene numbering intact
    reads = [b"GGGGGGGG", b"ACGTACGTA"]
    assert_same(gpu_hits(eng, c, reads, targets, False), as_arr(orc.match_direct(reads, targets, c)))


@pytest.mark.parametrize("maxlen", [31, 48, 100, 109, 110, 112, 113, 150, 180, 189, 190, 200, 250, 400])
def test_read_length_strides(eng, maxlen):
    """Record strides 4/8/12/16 words (compiled) and the runtime-stride kernel; with Windows 0,11 the
    lengths up to 109 fit 120 bases of context, 110..189 the wide buckets' 200, longer reads take the
    two-kernel path."""
    rng = random.Random(maxlen)
    targets = [rand_seq(rng, rng.randint(maxlen, 3 * maxlen), b"ACGT") for _ in range(12)]
    targets += [mutate(rng, t, 0.02, b"ACGT") for t in targets[:5]]
    reads = set()
    for _ in range(60):
        t = rng.choice(targets)
        L = rng.randint(max(20, maxlen // 2), maxlen)
        p = rng.choice([0, len(t) - L, rng.randint(0, len(t) - L)])
        reads.add(mutate(rng, t[p:p + L], rng.choice([0, 0.02, 0.05]), b"ACGT"))
    reads.add(rand_seq(rng, maxlen, b"ACGT"))
    reads = sorted(reads)
    c = orc.Config(Windows=[0, 11], WindowWidth=10, PMatch=0.9, MinDinuc=3, MaxReadLength=maxlen, MMTol=2)
    assert_same(gpu_hits(eng, c, reads, targets, False), as_arr(orc.match_direct(reads, targets, c)))
    if eng.index_mode in ("auto", "dma"):
        assert eng.stats()["index_kind"] == (1 if maxlen <= 109 else 2 if maxlen <= 189 else 0)


def test_wide_windows(eng):
    """WindowWidth > 16 uses the hashed index; > 32 folds several key words."""
    rng = random.Random(77)
    targets = [rand_seq(rng, 400, b"ACGT") for _ in range(10)]
    reads = sorted({mutate(rng, t[p:p + 120], 0.01, b"ACGT") for t in targets for p in (0, 37, 280)})
    for ww in (17, 20, 31, 32, 33, 40, 64):
        c = orc.Config(Windows=[0, 45], WindowWidth=ww, PMatch=0.95, MaxReadLength=120)
        assert_same(gpu_hits(eng, c, reads, targets, False), as_arr(orc.match_direct(reads, targets, c)))


WIDE_CASES = [
    # windows, ww, read length, pmatch, x in the reads, n_targets, n_reads
    ((0, 20, 40), 15, 100, 0.97, 0.0, 8000, 100000),   # BASELINE configs[4]'s windows on a database that fits
    ((0, 20, 40), 11, 100, 0.95, 0.0, 3000, 60000),    # 11-mers: 0.7 entries per key -- many buckets beyond two entries
    ((0, 20), 15, 150, 0.96, 0.0, 6000, 80000),        # 150-bp reads, records of 12 words
    ((0, 25), 17, 175, 0.95, 0.0, 4000, 50000),        # 175-bp reads: records of 12 words, hashed table, 200 bases exactly
    ((0, 20, 40, 60), 12, 140, 0.975, 0.003, 3000, 60000),  # four windows, X in the reads (budget 3 < the 4 X a record lists)
    ((0, 20), 15, 180, 0.97, 0.0, 3000, 40000),        # records of 16 words
]


@pytest.mark.parametrize("windows,ww,L,pmatch,xrate,nt,nr", WIDE_CASES)
def test_wide_context_buckets_against_literal_oracle(eng, windows, ww, L, pmatch, xrate, nt, nr):
    """Runs beyond 120 bases of context (three windows, 150-bp reads ...): on wide context buckets
    where k_match_t runs (index_kind 2), on the two-kernel path elsewhere -- the same tuples as the
    reference-shaped C++ oracle, union and best+MMTol selection."""
    from muscato_amd import sorted_hits
    reads, targets = synthetic_medium(4000 + ww + L, nt, nr, L=L)
    if xrate:
        rng = np.random.default_rng(L)
        R = np.frombuffer(b"".join(reads), dtype=np.uint8).reshape(-1, L).copy()
        R[rng.random(R.shape) < xrate] = ord("X")
        reads = sorted({bytes(r) for r in R})
    c = orc.Config(Windows=list(windows), WindowWidth=ww, PMatch=pmatch, MinDinuc=3, MaxReadLength=L,
                   MaxMatches=1000000, MMTol=1)
    rbuf, roff = literal.concat(reads)
    gbuf, goff = literal.concat(targets)
    exp, _, _ = literal.match_arrays(rbuf, roff, gbuf, goff,
                                     literal.make_params(c, bloom_size=256_000_000, num_hash=8, nthreads=8))
    got = gpu_hits(eng, c, reads, targets, False)
    st = eng.stats()
    if eng.index_mode in ("auto", "dma"):
        assert st["index_kind"] == 2
        if ww <= 12:
            assert st["n_overflow_entries"] > 1000
    assert len(got) > nr // 4
    assert_same(got, exp)
    best = sorted_hits(eng.match(to_cfg(c), apply_mmtol=True))
    assert_same(best, as_arr(orc.best_filter([tuple(int(x) for x in r) for r in exp], c.MMTol)))


def synthetic_medium(seed, n_targets, n_reads, tlen=1000, L=100, xrate=0.0):
    rng = np.random.default_rng(seed)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    T = bases[rng.integers(0, 4, size=(n_targets, tlen))]
    ncopy = n_targets // 5
    src = rng.integers(0, n_targets - ncopy, size=ncopy)
    T[n_targets - ncopy:] = T[src]
    sub = rng.random((ncopy, tlen)) < 0.02
    T[n_targets - ncopy:][sub] = bases[rng.integers(0, 4, size=int(sub.sum()))]
    if xrate:
        T[rng.random(T.shape) < xrate] = ord("X")
    g = rng.integers(0, n_targets, size=n_reads)
    p = rng.integers(0, tlen - L + 1, size=n_reads)
    p[rng.random(n_reads) < 0.01] = 0
    p[rng.random(n_reads) < 0.01] = tlen - L
    R = T[g[:, None], p[:, None] + np.arange(L)[None, :]].copy()
    sub = rng.random(R.shape) < 0.01
    R[sub] = bases[rng.integers(0, 4, size=int(sub.sum()))]
    nrand = n_reads // 5
    R[:nrand] = bases[rng.integers(0, 4, size=(nrand, L))]
    if xrate:
        R[rng.random(R.shape) < xrate] = ord("X")
    reads = sorted({bytes(r) for r in R})
    targets = [bytes(t) for t in T]
    return reads, targets


@pytest.mark.parametrize("xrate", [0.0, 0.001])
def test_medium_synthetic_against_literal_oracle(eng, xrate):
    """20k reads x 2k targets of 1000 bp, the BASELINE config-2 parameters, against the
    reference-shaped C++ oracle (Bloom + rolling hash screen, sorted merge-join confirm)."""
    reads, targets = synthetic_medium(11, 2000, 20000, xrate=xrate)
    c = orc.Config(Windows=[0, 20], WindowWidth=15, PMatch=0.97, MinDinuc=5, MaxReadLength=100,
                   MaxMatches=1000000, MMTol=0)
    rbuf, roff = literal.concat(reads)
    gbuf, goff = literal.concat(targets)
    exp, _, _ = literal.match_arrays(rbuf, roff, gbuf, goff,
                                     literal.make_params(c, bloom_size=64_000_000, num_hash=8, nthreads=8))
    got = gpu_hits(eng, c, reads, targets, False)
    assert len(got) > 10000
    assert_same(got, exp)
    st = eng.stats()
    assert st["n_pairs"] >= st["n_accepted"] >= len(got) > 0
    assert st["n_reads"] == len(reads)


def test_specialised_instance_is_guarded(eng):
    """BASELINE config 2's shape -- WindowWidth 15, Windows 0,20, MinDinuc 5, 100-base reads, i.e. every quantity the
    geometry-specialised kernel instance (SpecGeom<1>: BASELINE configs 3 / 4) turns into a constant -- on a database
    small enough to get a HASHED table, which that instance was not built for (it takes the key for the bucket: the
    r03 fault, gpurun_out/var.err, was a development build of that kind reading past cfg2's 2^27-bucket table).  The
    instance is in the library; the host's guard (spec_geom_matches: idx_direct / idx_bits are part of the comparison)
    must pick the GENERAL instance, and the tuples must equal the CPU port's.  (The whole-cfg3 test asserts the other
    direction: a direct 2^30-bucket table with the same geometry runs the specialised instance.)"""
    reads, targets = synthetic_medium(12, 3000, 30000)
    c = orc.Config(Windows=[0, 20], WindowWidth=15, PMatch=0.97, MinDinuc=5, MaxReadLength=100,
                   MaxMatches=1000000, MMTol=0)
    rbuf, roff = literal.concat(reads)
    gbuf, goff = literal.concat(targets)
    exp, _, _ = literal.match_arrays(rbuf, roff, gbuf, goff,
                                     literal.make_params(c, bloom_size=64_000_000, num_hash=8, nthreads=8))
    got = gpu_hits(eng, c, reads, targets, False)
    st = eng.stats()
    # 2 = k_match_t general, 4 = k_match_g general (3 / 5 would be the specialised ones), 0 = two-kernel path
    assert st["match_variant"] == {"auto": 2, "dma": 4}.get(eng.index_mode, 0), st["match_variant"]
    assert len(got) > 15000
    assert_same(got, exp)


MEDIUM_MATRIX = [
    # windows, ww, pmatch, mindinuc, mmtol, xrate, n_targets, n_reads
    ((0, 20, 40), 15, 0.97, 3, 3, 0.0, 20000, 200000),      # BASELINE configs[4] shape: odd window count
    ((0, 20, 40), 15, 0.95, 5, 1, 0.002, 8000, 120000),     # the same with X on both sides (mask planes)
    ((0, 30), 17, 0.96, 0, 0, 0.0, 8000, 120000),           # 17-mers: hashed index (34-bit keys)
    ((5, 25, 45, 65), 12, 0.94, 1, 2, 0.001, 4000, 80000),  # four windows, none at 0, 12-mers (dense buckets)
    ((0, 1), 15, 0.93, 0, 0, 0.0, 4000, 60000),             # overlapping windows: most placements found twice
]


@pytest.mark.parametrize("windows,ww,pmatch,mindinuc,mmtol,xrate,nt,nr", MEDIUM_MATRIX)
def test_medium_matrix_against_literal_oracle(eng, windows, ww, pmatch, mindinuc, mmtol, xrate, nt, nr):
    """Tens of thousands of reads per configuration against the reference-shaped C++ oracle:
    both the union over windows and the best+MMTol selection, plus the MaxMatches verdict."""
    from muscato_amd import sorted_hits
    reads, targets = synthetic_medium(1000 + ww + len(windows), nt, nr, xrate=xrate)
    c = orc.Config(Windows=list(windows), WindowWidth=ww, PMatch=pmatch, MinDinuc=mindinuc, MaxReadLength=100,
                   MaxMatches=1000000, MMTol=mmtol)
    rbuf, roff = literal.concat(reads)
    gbuf, goff = literal.concat(targets)
    exp, _, _ = literal.match_arrays(rbuf, roff, gbuf, goff,
                                     literal.make_params(c, bloom_size=256_000_000, num_hash=8, nthreads=8))
    got = gpu_hits(eng, c, reads, targets, False)
    assert len(got) > nr // 4
    assert_same(got, exp)
    assert eng.stats()["n_overflow_blocks"] == 0
    best = sorted_hits(eng.match(to_cfg(c), apply_mmtol=True))
    assert_same(best, as_arr(orc.best_filter([tuple(int(x) for x in r) for r in exp], mmtol)))


def test_packed_tuples_roundtrip(eng):
    """musc_hits_copy_packed / musc_hits_unpack: 8-byte words whose numeric order is the tuple
    order, exact round trip, and a loud failure when a field does not fit its width."""
    import ctypes
    from muscato_amd import sorted_hits
    ocfg, reads, targets = make_case(77)
    exp = gpu_hits(eng, ocfg, reads, targets, False)
    n = len(exp)
    assert n > 10
    bits = [max(1, (len(reads) + 999).bit_length()), max(1, len(targets).bit_length()),
            max(1, max(len(t) for t in targets).bit_length()), 8]
    words = np.zeros(n, dtype=np.uint64)
    eng.hits_to_packed(words.ctypes.data, n, False, bits, read_base=1000)
    back = np.zeros((n, 4), dtype=np.uint32)
    eng.unpack_hits(words.ctypes.data, n, False, bits, back.ctypes.data)
    back[:, 0] -= 1000
    assert (sorted_hits(back) == exp).all()
    # numeric order of the words == lexicographic order of the tuples
    order = np.argsort(words, kind="stable")
    assert (back[order] == exp).all()
    # numpy view of the layout
    w = words.astype(np.uint64)
    assert ((w & np.uint64(255)) == back[:, 3]).all()
    # device destinations (what the RCCL path uses); plain HIP runtime calls, no torch
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    dw, dh = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dw), n * 8) == 0 and hip.hipMalloc(ctypes.byref(dh), n * 16) == 0
    try:
        eng.hits_to_packed(dw.value, n, True, bits, read_base=1000)
        w2 = np.zeros(n, dtype=np.uint64)
        assert hip.hipMemcpy(w2.ctypes.data, dw, n * 8, 2) == 0   # hipMemcpyDeviceToHost
        assert (w2 == words).all()
        eng.unpack_hits(dw.value, n, True, bits, dh.value)
        h2 = np.zeros((n, 4), dtype=np.uint32)
        assert hip.hipMemcpy(h2.ctypes.data, dh, n * 16, 2) == 0
        assert (h2 == back + np.array([1000, 0, 0, 0], dtype=np.uint32)).all()
    finally:
        hip.hipFree(dw)
        hip.hipFree(dh)
    with pytest.raises(RuntimeError, match="does not fit"):
        eng.hits_to_packed(words.ctypes.data, n, False, [bits[0], bits[1], 1, 8])
    with pytest.raises(RuntimeError, match="64 bits"):
        eng.hits_to_packed(words.ctypes.data, n, False, [32, 32, 32, 8])


def test_compact_tuples(eng):
    """musc_hits_copy_compact: one u32 word per tuple + one count byte per read (the hit list is
    read-major), to host and to device memory; loud failures when a field does not fit, when the
    widths exceed 32 bits, and when a read has more than 255 tuples."""
    import ctypes
    ocfg, reads, targets = make_case(77)
    exp = gpu_hits(eng, ocfg, reads, targets, False)
    n, nr = len(exp), len(reads)
    bits = [max(1, len(targets).bit_length()), max(1, max(len(t) for t in targets).bit_length()), 8]
    assert sum(bits) <= 32
    words = np.zeros(n, dtype=np.uint32)
    counts = np.full(nr, 77, dtype=np.uint8)
    eng.hits_to_compact(words.ctypes.data, n, counts.ctypes.data, nr, False, bits)
    assert int(counts.sum()) == n and (counts == np.bincount(exp[:, 0], minlength=nr)).all()
    rd = np.repeat(np.arange(nr, dtype=np.uint32), counts)
    w = words.astype(np.uint64)
    back = np.stack([rd, (w >> np.uint64(bits[1] + bits[2])).astype(np.uint32),
                     ((w >> np.uint64(bits[2])) & np.uint64((1 << bits[1]) - 1)).astype(np.uint32),
                     (w & np.uint64((1 << bits[2]) - 1)).astype(np.uint32)], axis=1)
    from muscato_amd import sorted_hits
    assert (sorted_hits(back) == exp).all()
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    dw, dc = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dw), n * 4 + 16) == 0 and hip.hipMalloc(ctypes.byref(dc), nr + 16) == 0
    try:
        eng.hits_to_compact(dw.value, n, dc.value, nr, True, bits)
        w2, c2 = np.zeros(n, dtype=np.uint32), np.zeros(nr, dtype=np.uint8)
        assert hip.hipMemcpy(w2.ctypes.data, dw, n * 4, 2) == 0 and hip.hipMemcpy(c2.ctypes.data, dc, nr, 2) == 0
        assert (w2 == words).all() and (c2 == counts).all()
    finally:
        hip.hipFree(dw)
        hip.hipFree(dc)
    with pytest.raises(RuntimeError, match="does not fit"):
        eng.hits_to_compact(words.ctypes.data, n, counts.ctypes.data, nr, False, [bits[0], 1, 8])
    with pytest.raises(RuntimeError, match="32 bits"):
        eng.hits_to_compact(words.ctypes.data, n, counts.ctypes.data, nr, False, [20, 10, 8])
    # one read matching 300 targets: more tuples than a count byte holds
    tg = [b"ACGTACGTAGGATCCATTGA" + rand_seq(random.Random(i), 12, b"ACGT") for i in range(300)]
    c = orc.Config(Windows=[0], WindowWidth=10, PMatch=1.0, MaxReadLength=20, MaxMatches=100000)
    got = gpu_hits(eng, c, [b"ACGTACGTAGGATCCATTGA"], tg, False)
    assert len(got) == 300
    w3, c3 = np.zeros(300, dtype=np.uint32), np.zeros(1, dtype=np.uint8)
    with pytest.raises(RuntimeError, match="255 tuples"):
        eng.hits_to_compact(w3.ctypes.data, 300, c3.ctypes.data, 1, False, [12, 8, 4])


def test_sixteen_windows(eng):
    """The maximum number of windows (eight chunks of two in k_screen, a 16 KB window-counter
    block in k_confirm), overlapping and at odd offsets, with a tight MaxMatches check."""
    from muscato_amd import sorted_hits
    reads, targets = synthetic_medium(77, 1500, 20000, tlen=600, L=100)
    wins = [0, 3, 7, 11, 16, 22, 29, 31, 37, 41, 48, 53, 59, 66, 73, 80]
    c = orc.Config(Windows=wins, WindowWidth=12, PMatch=0.95, MinDinuc=2, MaxReadLength=100, MaxMatches=1000000, MMTol=1)
    rbuf, roff = literal.concat(reads)
    gbuf, goff = literal.concat(targets)
    exp, _, _ = literal.match_arrays(rbuf, roff, gbuf, goff,
                                     literal.make_params(c, bloom_size=64_000_000, num_hash=6, nthreads=8))
    got = gpu_hits(eng, c, reads, targets, False)
    assert len(got) > 5000
    assert_same(got, exp)
    assert eng.stats()["n_overflow_blocks"] == 0
    best = sorted_hits(eng.match(to_cfg(c), apply_mmtol=True))
    assert_same(best, as_arr(orc.best_filter([tuple(int(x) for x in r) for r in exp], 1)))
    with pytest.raises(RuntimeError, match="at most 16 windows"):
        c17 = orc.Config(Windows=wins + [85], WindowWidth=12, PMatch=0.95, MaxReadLength=100)
        gpu_hits(eng, c17, reads[:10], targets[:10], False)


def test_low_complexity_stress(eng):
    """Poly-A-like targets and reads: every window key occurs at almost every target position,
    so one tile of 100 reads carries millions of descriptors (overflow lists of ~30 000 entries
    per bucket, result words spilling past the LDS buffer, region growth + batch repeat).  Tuples
    against the literal oracle; MaxMatches small enough to be reached must be reported."""
    from muscato_amd import sorted_hits
    rng = np.random.default_rng(5)

    def lowc(n, rate):
        a = np.full(n, ord("A"), dtype=np.uint8)
        m = rng.random(n) < rate
        a[m] = np.frombuffer(b"CGT", dtype=np.uint8)[rng.integers(0, 3, size=int(m.sum()))]
        return bytes(a)
    targets = [lowc(1000, 0.01) for _ in range(32)]
    reads = sorted({lowc(int(rng.integers(30, 61)), 0.02) for _ in range(100)})
    c = orc.Config(Windows=[0, 12], WindowWidth=8, PMatch=0.9, MinDinuc=0, MaxReadLength=60,
                   MaxMatches=1_000_000_000, MMTol=0)
    rbuf, roff = literal.concat(reads)
    gbuf, goff = literal.concat(targets)
    exp, _, _ = literal.match_arrays(rbuf, roff, gbuf, goff,
                                     literal.make_params(c, bloom_size=64_000_000, num_hash=6, nthreads=8))
    got = gpu_hits(eng, c, reads, targets, False)
    st = eng.stats()
    assert st["index_kind"] == {"classic": 0, "lines": 3}.get(eng.index_mode, 1)
    assert (st["n_descriptors"] if st["index_kind"] in (0, 3) else st["n_pairs"]) > 1_000_000 and len(got) > 500_000
    assert_same(got, exp)
    assert st["n_overflow_blocks"] == 0
    best = sorted_hits(eng.match(to_cfg(c), apply_mmtol=True))
    assert_same(best, as_arr(orc.best_filter([tuple(int(x) for x in r) for r in exp], 0)))
    c.MaxMatches = 10000
    gpu_hits(eng, c, reads, targets, False)
    assert eng.stats()["n_overflow_blocks"] >= 1 and len(eng.overflow_probes()) >= 1


def test_stats_and_repeat_calls_are_stable(eng):
    from muscato_amd import sorted_hits
    ocfg, reads, targets = make_case(4)
    a = gpu_hits(eng, ocfg, reads, targets, False)
    b = sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=False))
    assert_same(a, b)
    st = eng.stats()
    assert st["n_hits"] == len(a) and st["ms_total"] >= 0


def test_repeated_passes_skip_sizing_and_stay_identical(eng):
    """A pass over the same reads, database and parameters as the previous one runs without
    host round trips (buffers already sized); it must give the same tuples and counters, and
    new reads or parameters must take the careful path again."""
    from muscato_amd import sorted_hits
    keys = ("n_read_windows", "n_candidates", "n_pairs", "n_accepted", "n_hits", "n_overflow_blocks")
    for seed in (3, 8, 21, 77):
        ocfg, reads, targets = make_case(seed)
        exp = as_arr(orc.match_direct(reads, targets, ocfg))
        first = gpu_hits(eng, ocfg, reads, targets, False)
        st0 = eng.stats()
        assert_same(first, exp)
        for _ in range(3):
            assert_same(sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=False)), exp)
            st = eng.stats()
            assert [st[k] for k in keys] == [st0[k] for k in keys]
        best = as_arr(orc.best_filter(orc.match_direct(reads, targets, ocfg), ocfg.MMTol))
        assert_same(sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=True)), best)
        eng.load_reads(reads[: max(1, len(reads) // 2)])
        half = as_arr(orc.match_direct(reads[: max(1, len(reads) // 2)], targets, ocfg))
        assert_same(sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=False)), half)


def test_maxmatches_overflow_is_detected(eng):
    """cmd/muscato_confirm/main.go:233-242, 424-448: a (window,key) block with more than
    MaxMatches accepted pairs is truncated order-dependently by the reference.  The GPU path keeps
    every tuple and must say so: n_overflow_blocks > 0 exactly when a block overflows."""
    key = b"ACGT"
    targets = [key + bytes([b"ACGT"[(i >> s) & 3] for s in (0, 2)]) + b"AAAA" for i in range(12)]
    reads = [key + b"AA"]
    c = orc.Config(Windows=[0], WindowWidth=4, PMatch=0.5, MaxReadLength=50, MaxMatches=5)
    got = gpu_hits(eng, c, reads, targets, False)
    assert_same(got, as_arr(orc.match_direct(reads, targets, c, check_overflow=False)))
    assert eng.stats()["n_overflow_blocks"] >= 1
    probes = eng.overflow_probes()
    assert probes.tolist() == [[0, 0]]  # read 0 through window 0 sits in the overflowing block
    lit = literal.match_literal(reads, targets, c)
    assert set(map(tuple, lit)) < set(map(tuple, got.tolist()))   # the reference keeps a strict subset
    c.MaxMatches = 12                                              # exactly at the limit: no truncation
    got = gpu_hits(eng, c, reads, targets, False)
    assert eng.stats()["n_overflow_blocks"] == 0 and len(eng.overflow_probes()) == 0
    assert_same(got, np.array(literal.match_literal(reads, targets, c), dtype=np.uint32).reshape(-1, 4))
    # two windows over the same placements: each window's block counts its own acceptances
    c = orc.Config(Windows=[0, 1], WindowWidth=4, PMatch=0.5, MaxReadLength=50, MaxMatches=11)
    gpu_hits(eng, c, reads, targets, False)
    assert eng.stats()["n_overflow_blocks"] >= 1
    assert eng.overflow_probes().tolist() == [[0, 0]]  # window 1 (CGTA) only matches targets 0, 4, 8
    # both windows match every target: one descriptor stands for both windows of a placement and
    # both blocks must still count it
    targets = [b"ACGTA" + bytes([b"ACGT"[(i >> s) & 3] for s in (0, 2)]) + b"AAAA" for i in range(12)]
    reads = [b"ACGTAAA"]
    got = gpu_hits(eng, c, reads, targets, False)
    assert_same(got, as_arr(orc.match_direct(reads, targets, c, check_overflow=False)))
    assert len(got) == 12 and eng.stats()["n_overflow_blocks"] >= 1
    assert sorted(eng.overflow_probes().tolist()) == [[0, 0], [0, 1]]
    c.MaxMatches = 12
    gpu_hits(eng, c, reads, targets, False)
    assert eng.stats()["n_overflow_blocks"] == 0 and len(eng.overflow_probes()) == 0
    # three windows (the k_match_t instances that keep a wave-tile's bucket numbers and own counts in LDS, r04): windows 0
    # and 1 match every target, window 2 (GTAA) only targets 0, 4, 8 -- each block counts its own acceptances
    c = orc.Config(Windows=[0, 1, 2], WindowWidth=4, PMatch=0.5, MaxReadLength=50, MaxMatches=11)
    got = gpu_hits(eng, c, reads, targets, False)
    assert_same(got, as_arr(orc.match_direct(reads, targets, c, check_overflow=False)))
    assert len(got) == 12 and eng.stats()["n_overflow_blocks"] >= 1
    assert sorted(eng.overflow_probes().tolist()) == [[0, 0], [0, 1]]
    c.MaxMatches = 2   # now window 2's block of three overflows as well
    gpu_hits(eng, c, reads, targets, False)
    assert sorted(eng.overflow_probes().tolist()) == [[0, 0], [0, 1], [0, 2]]
    c.MaxMatches = 12
    gpu_hits(eng, c, reads, targets, False)
    assert eng.stats()["n_overflow_blocks"] == 0 and len(eng.overflow_probes()) == 0


def test_no_overflow_reported_on_ordinary_cases(eng):
    for seed in range(0, 40, 3):
        ocfg, reads, targets = make_case(seed)
        gpu_hits(eng, ocfg, reads, targets, True)
        assert eng.stats()["n_overflow_blocks"] == 0


def test_block_screening_falls_back_to_exact_counters(eng):
    """MaxMatches = 20000 gives a screening threshold of 2 accepted pairs per workgroup sketch
    cell: any multi-mapping read trips it and the pass is repeated with exact per-block counters;
    the tuples and the verdict (no overflow) must not change."""
    for seed in (2, 6, 11, 17):
        ocfg, reads, targets = make_case(seed)
        ocfg.MaxMatches = 20000
        full = orc.match_direct(reads, targets, ocfg)
        assert_same(gpu_hits(eng, ocfg, reads, targets, False), as_arr(full))
        assert eng.stats()["n_overflow_blocks"] == 0


def test_index_selection(monkeypatch):
    """Which index a run gets (ensure_index): context buckets + the fused kernel when every read
    fits 120 bases (index_kind 1) or 200 bases (2: wide buckets) of context around each of at most
    four windows (a database with X included: test_database_with_x_on_context_buckets); the
    window-start buckets and k_screen -> k_confirm otherwise -- with identical tuples either way."""
    from muscato_amd import Config, Engine, sorted_hits
    monkeypatch.delenv("MUSC_INDEX", raising=False)
    monkeypatch.delenv("MUSC_MATCH", raising=False)
    rng = random.Random(2)
    targets = [rand_seq(rng, 400, b"ACGT") for _ in range(40)]

    def reads_of(L, n=200, alphabet=b"ACGT"):
        out = set()
        for _ in range(n):
            t = rng.choice(targets)
            p = rng.randint(0, len(t) - L)
            out.add(mutate(rng, t[p:p + L], 0.02, alphabet))
        return sorted(out)

    cases = [
        ("two windows, 100 bp: exactly 120 bases of context", [0, 20], 100, b"ACGT", 1),
        ("one base more: wide context buckets (200 bases, two entries per line)", [0, 21], 100, b"ACGT", 2),
        ("BASELINE configs[4]'s windows", [0, 20, 40], 100, b"ACGT", 2),
        ("150-bp reads", [0, 20], 150, b"ACGT", 2),
        ("exactly 200 bases of context", [0, 20], 180, b"ACGT", 2),
        ("one base too many for those too", [0, 20], 181, b"ACGT", 0),
        ("a distant window with short reads", [0, 100], 40, b"ACGT", 2),
        ("four windows, 90 bp", [0, 10, 20, 30], 90, b"ACGT", 1),
        ("five windows", [0, 5, 10, 15, 20], 90, b"ACGT", 0),
        # (reads with X at PMatch 0.97: a read with more X than its xpos word lists -- four, on wide
        # buckets three -- then exceeds its budget of three mismatches and takes no part)
        ("reads with X, database without: k_match_t lists a read's X in its xpos word", [0, 20], 100, b"ACGTX", 1),
        ("reads with X, three windows", [0, 10, 20], 100, b"ACGTX", 1),
        ("reads with X on wide buckets", [0, 20, 40], 100, b"ACGTX", 2),
        ("long reads", [0, 20], 250, b"ACGT", 0),
    ]
    with Engine(0) as eng:
        eng.load_targets(targets)
        for what, wins, L, alphabet, kind in cases:
            reads = reads_of(L, alphabet=alphabet)
            ocfg = orc.Config(Windows=wins, WindowWidth=12, PMatch=0.97 if b"X" in alphabet else 0.9, MinDinuc=2, MaxReadLength=L,
                              MaxMatches=100000, MMTol=1)
            eng.load_reads(reads)
            got = sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=False))
            assert eng.stats()["index_kind"] == kind, what
            assert_same(got, as_arr(orc.match_direct(reads, targets, ocfg)))
        # a database with X keeps its context buckets (entries whose context touches an X are flagged and
        # compared through the mask plane); MUSC_NO_X_CONTEXT=1 sends it to the two-kernel path as before
        xt = [t[:50] + b"X" + t[51:] for t in targets[:3]] + targets[3:]
        eng.load_targets(xt)
        reads = reads_of(100)
        ocfg = orc.Config(Windows=[0, 20], WindowWidth=12, PMatch=0.9, MinDinuc=2, MaxReadLength=100, MaxMatches=100000)
        eng.load_reads(reads)
        got = sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=False))
        assert eng.stats()["index_kind"] == 1
        assert_same(got, as_arr(orc.match_direct(reads, xt, ocfg)))
        monkeypatch.setenv("MUSC_NO_X_CONTEXT", "1")
        eng.reload_env()  # (the library reads its MUSC_* knobs once per context)
        got = sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=False))
        assert eng.stats()["index_kind"] == 0
        assert_same(got, as_arr(orc.match_direct(reads, xt, ocfg)))
        monkeypatch.delenv("MUSC_NO_X_CONTEXT")
        eng.load_targets(targets)
        # forced
        monkeypatch.setenv("MUSC_INDEX", "classic")
        eng.reload_env()
        reads = reads_of(100)
        ocfg = orc.Config(Windows=[0, 20], WindowWidth=12, PMatch=0.9, MinDinuc=2, MaxReadLength=100, MaxMatches=100000)
        eng.load_reads(reads)
        got = sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=False))
        assert eng.stats()["index_kind"] == 0
        assert_same(got, as_arr(orc.match_direct(reads, targets, ocfg)))
        # musc_db_build_index_for builds what the match then uses (no rebuild: the timing stays)
        monkeypatch.delenv("MUSC_INDEX", raising=False)
        eng.reload_env()
        eng.build_index_for(to_cfg(ocfg), 100)
        ms = eng.stats()["ms_index_build"]
        got = sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=False))
        assert eng.stats()["index_kind"] == 1 and eng.stats()["ms_index_build"] == ms
        assert_same(got, as_arr(orc.match_direct(reads, targets, ocfg)))


@pytest.mark.gpu
def test_three_window_block_accounting_agrees_across_paths(monkeypatch):
    """MaxMatches accounting with three windows and many reads per (window, key) block: the fused kernel's
    three-window instances keep a wave-tile's bucket numbers and own accepted-pair counts in LDS (r04), the
    two-kernel path counts in k_confirm -- both must name the same overflowing probes, none when MaxMatches
    is large, and return the oracle's tuples either way."""
    from muscato_amd import Engine, sorted_hits
    monkeypatch.delenv("MUSC_INDEX", raising=False)
    monkeypatch.delenv("MUSC_MATCH", raising=False)
    rng = random.Random(31)
    motif = rand_seq(rng, 90, b"ACGT")
    targets = [rand_seq(rng, 300, b"ACGT") for _ in range(80)]
    for i in range(0, 40):   # the motif sits in forty targets: every read from it has forty placements
        p = rng.randint(0, 200)
        targets[i] = targets[i][:p] + motif + targets[i][p + 90:]
    reads = set()
    for _ in range(400):
        if rng.random() < 0.7:
            p = rng.randint(0, 30)
            reads.add(mutate(rng, motif[p:p + 60], 0.01, b"ACGT"))
        else:
            t = rng.choice(targets[40:])
            p = rng.randint(0, 240)
            reads.add(mutate(rng, t[p:p + 60], 0.01, b"ACGT"))
    reads = sorted(reads)
    ocfg = orc.Config(Windows=[0, 10, 20], WindowWidth=10, PMatch=0.9, MinDinuc=2, MaxReadLength=60, MaxMatches=25, MMTol=2)
    full = as_arr(orc.match_direct(reads, targets, ocfg, check_overflow=False))
    seen = {}
    with Engine(0) as eng:
        eng.load_targets(targets)
        eng.load_reads(reads)
        for mode in ("auto", "classic"):
            if mode == "classic":
                monkeypatch.setenv("MUSC_INDEX", "classic")
            eng.reload_env()
            got = sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=False))
            assert eng.stats()["index_kind"] == (1 if mode == "auto" else 0)
            assert_same(got, full)
            assert eng.stats()["n_overflow_blocks"] >= 1
            seen[mode] = sorted(map(tuple, eng.overflow_probes().tolist()))
        assert seen["auto"] == seen["classic"] and len(seen["auto"]) > 10
        assert {k for _, k in seen["auto"]} == {0, 1, 2}   # every window has overflowing blocks
        big = orc.Config(Windows=[0, 10, 20], WindowWidth=10, PMatch=0.9, MinDinuc=2, MaxReadLength=60, MaxMatches=1000000, MMTol=2)
        for mode in ("classic", "auto"):
            if mode == "auto":
                monkeypatch.delenv("MUSC_INDEX")
            eng.reload_env()
            got = sorted_hits(eng.match(to_cfg(big), apply_mmtol=False))
            assert_same(got, full)
            assert eng.stats()["n_overflow_blocks"] == 0 and len(eng.overflow_probes()) == 0


@pytest.mark.gpu
def test_database_with_x_on_context_buckets(monkeypatch):
    """A database with X (N in the FASTA) stays on context buckets: a window that holds an X is not
    indexed, an entry whose context touches one is flagged and k_match_t<.., XM = 2> reads the target
    span's mask plane for it -- X == X is a match, X against a base a mismatch
    (cmd/muscato_confirm/main.go:151-159).  Reads with X keep the run there while every X is listed in
    the read's xpos word and sits outside the run's windows; otherwise the two-kernel path takes over.
    Every configuration against the direct oracle, the index kind asserted."""
    from muscato_amd import Engine, sorted_hits
    monkeypatch.delenv("MUSC_INDEX", raising=False)
    monkeypatch.delenv("MUSC_MATCH", raising=False)
    monkeypatch.delenv("MUSC_NO_X_CONTEXT", raising=False)

    def sprinkle(rng, t, rate, run):
        """X at `rate` per base, as runs of `run` bases"""
        b = bytearray(t)
        i = 0
        while i < len(b):
            if rng.random() < rate:
                for j in range(i, min(len(b), i + run)):
                    b[j] = ord("X")
                i += run
            i += 1
        return bytes(b)

    def reads_from(rng, targets, L, n, sub, keep_x, xfree_windows=None, ww=0):
        out = set()
        for _ in range(n):
            t = rng.choice(targets)
            if len(t) < L:
                continue
            u = rng.random()
            p = 0 if u < 0.15 else (len(t) - L if u > 0.9 else rng.randint(0, len(t) - L))
            r = bytearray(mutate(rng, t[p:p + L], sub, b"ACGT"))
            for i in range(L):
                x = t[p + i] == ord("X")
                if x and keep_x:
                    inwin = xfree_windows is not None and any(q <= i < q + ww for q in xfree_windows)
                    r[i] = ord("X") if not inwin else ord("A")
                elif x:
                    r[i] = rng.choice(b"ACGT")
            out.add(bytes(r))
        return sorted(out)

    # (what, windows, ww, read length, X rate, X run length, reads keep the target's X?, expected kind)
    grid = [
        ("isolated X, X-free reads", [0, 20], 12, 100, 0.004, 1, False, 1),
        ("runs of 30 X, X-free reads", [0, 20], 12, 100, 0.002, 30, False, 1),
        ("dense X, one window", [5], 10, 80, 0.02, 2, False, 1),
        ("three windows at 100 bp: wide buckets", [0, 20, 40], 12, 100, 0.004, 3, False, 2),
        ("150-bp reads: wide buckets", [0, 25], 14, 150, 0.003, 5, False, 2),
        ("four windows, 15-mers", [0, 10, 20, 30], 15, 90, 0.004, 1, False, 1),
        ("reads carry the target's X outside their windows", [0, 20], 12, 100, 0.004, 1, "outside", 1),
        ("the same on wide buckets", [0, 20, 40], 12, 100, 0.004, 2, "outside", 2),
        ("reads carry the target's X anywhere: some window holds one", [0, 20], 12, 100, 0.01, 1, True, 0),
    ]
    with Engine(0) as eng:
        for gi, (what, wins, ww, L, rate, run, keep, kind) in enumerate(grid):
            rng = random.Random(100 + gi)
            base = [rand_seq(rng, rng.randint(L, 500), b"ACGT") for _ in range(30)]
            base += [mutate(rng, rng.choice(base), 0.02, b"ACGT") for _ in range(15)]  # multi-map families
            targets = [sprinkle(rng, t, rate, run) for t in base]
            targets[0] = b"X" * 3 + targets[0][3:]                    # X at a target's start and end
            targets[1] = targets[1][:-2] + b"XX"
            assert any(b"X" in t for t in targets)
            reads = reads_from(rng, targets, L, 600, 0.02, bool(keep), wins if keep == "outside" else None, ww)
            if keep == "outside":
                reads = [r for r in reads if r.count(b"X") <= 3]   # what an xpos word lists on either bucket width
            total = covers = 0
            for pm, mmtol in ((0.9, 1), (0.97, 0), (1.0, 0)):
                ocfg = orc.Config(Windows=wins, WindowWidth=ww, PMatch=pm, MinDinuc=2, MaxReadLength=L, MaxMatches=100000, MMTol=mmtol)
                full = orc.match_direct(reads, targets, ocfg)
                total += len(full)
                covers += sum(b"X" in targets[g][p:p + len(reads[r])] for r, g, p, _ in full)
                got = gpu_hits(eng, ocfg, reads, targets, False)
                assert eng.stats()["index_kind"] == kind, what
                assert_same(got, as_arr(full))
                got = sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=True))
                assert_same(got, as_arr(orc.best_filter(full, ocfg.MMTol)))
            if kind:
                # the flagged entries matter: accepted placements cover an X of their target
                assert total > 300 and covers > 20, what


@pytest.mark.gpu
def test_graph_replay_of_the_sized_pass(monkeypatch):
    """MUSC_GRAPH=1: from the second identical pass on, the context-bucket pass is one hipGraph
    launch (memsets, k_match, scan, k_compact_w, counter readback).  Tuples and counters must equal
    the launch-by-launch pass, and new reads, new parameters or a different tuple selection must
    drop the captured graph instead of replaying stale arguments."""
    from muscato_amd import Engine, sorted_hits
    monkeypatch.delenv("MUSC_INDEX", raising=False)
    keys = ("n_hits", "n_pairs", "n_candidates", "n_read_windows", "n_accepted", "n_overflow_blocks", "n_overflow_entries",
            "index_kind", "n_batches", "match_launches", "match_bytes")
    with Engine(0) as eng:
        for seed in (5, 6):
            rng = random.Random(seed)
            targets = [rand_seq(rng, 600, b"ACGT") for _ in range(60)]
            eng.load_targets(targets)
            for L, wins, mm in ((100, [0, 20], 1), (80, [0, 15, 30], 0)):
                reads = sorted({mutate(rng, (t := rng.choice(targets))[(p := rng.randint(0, 600 - L)):p + L], 0.03, b"ACGT")
                                for _ in range(3000)})
                ocfg = orc.Config(Windows=wins, WindowWidth=12, PMatch=0.9, MinDinuc=2, MaxReadLength=L, MaxMatches=100000, MMTol=mm)
                eng.load_reads(reads)
                for mode in (True, False):
                    monkeypatch.delenv("MUSC_GRAPH", raising=False)
                    eng.reload_env()
                    exp = sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=mode))
                    st0 = eng.stats()
                    assert st0["index_kind"] == 1
                    monkeypatch.setenv("MUSC_GRAPH", "1")
                    eng.reload_env()  # (the next pass sizes itself again; then capture, replay)
                    for rep in range(4):  # sizing pass, capture, replay, replay
                        got = sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=mode))
                        st = eng.stats()
                        assert (got == exp).all() and all(st[k] == st0[k] for k in keys), (seed, L, mode, rep)
                        assert st["ms_total"] > 0
                full = as_arr(orc.match_direct(reads, targets, ocfg))
                assert_same(exp, full)


def test_batches_of_heavy_tiles_move_their_tuples_in_the_next_launch(monkeypatch):
    """Families of near-identical targets: a read has dozens of tuples, a wave-tile thousands (copy
    loops past the first 64, candidate lists spilling past LDS), in many small batches.  From the
    second pass on k_match_t moves a batch's staged tuples from inside the next batch's launch
    (last batch: k_compact_w; a last batch with a smaller grid: k_compact_w for the one before as
    well).  Every pass must return the single-batch list of the two-kernel path, in read-major
    order, with and without best + MMTol."""
    from muscato_amd import Engine, sorted_hits
    rng = np.random.default_rng(77)
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    fam = bases[rng.integers(0, 4, size=(12, 400))]
    T = np.repeat(fam, 30, axis=0)  # 12 families x 30 copies
    sub = rng.random(T.shape) < 0.01
    T[sub] = bases[rng.integers(0, 4, size=int(sub.sum()))]
    targets = [bytes(t) for t in T]
    g = rng.integers(0, len(targets), size=30000)
    p = rng.integers(0, 400 - 80 + 1, size=30000)
    R = T[g[:, None], p[:, None] + np.arange(80)[None, :]].copy()
    sub = rng.random(R.shape) < 0.01
    R[sub] = bases[rng.integers(0, 4, size=int(sub.sum()))]
    reads = sorted({bytes(r) for r in R})
    ocfg = orc.Config(Windows=[0, 25], WindowWidth=12, PMatch=0.9, MinDinuc=2, MaxReadLength=80,
                      MaxMatches=1_000_000_000, MMTol=1)
    monkeypatch.delenv("MUSC_MATCH", raising=False)
    monkeypatch.setenv("MUSC_INDEX", "classic")
    with Engine(0) as ref:
        ref.load_targets(targets)
        ref.load_reads(reads)
        exp_all = sorted_hits(ref.match(to_cfg(ocfg), apply_mmtol=False))
        exp_best = sorted_hits(ref.match(to_cfg(ocfg), apply_mmtol=True))
    assert len(exp_all) > 20 * len(reads)
    monkeypatch.delenv("MUSC_INDEX", raising=False)
    # (the in-launch move is opt-in since r04 -- a k_compact_w per batch is the default -- and stays covered here, with
    # both fused kernels; the plain form runs in every other multi-batch test)
    for batch, fused, kern in (("4099", "1", ""), ("8192", "1", "dma"), ("7000", "1", ""), ("7000", "1", "dma"), ("4099", "0", "")):  # 7000: the last batch of 30 000 has a smaller grid
        monkeypatch.setenv("MUSC_BATCH_READS", batch)  # read at musc_init
        monkeypatch.setenv("MUSC_FUSED_COMPACT", fused)
        if kern:
            monkeypatch.setenv("MUSC_MATCH", kern)
        else:
            monkeypatch.delenv("MUSC_MATCH", raising=False)
        with Engine(0) as eng:
            eng.load_targets(targets)
            eng.load_reads(reads)
            for mode, exp in ((False, exp_all), (True, exp_best)):
                for rep in range(3):  # pass 0 sizes, passes 1-2 are sized (same parameters as the pass before)
                    raw = eng.match(to_cfg(ocfg), apply_mmtol=mode)
                    st = eng.stats()
                    assert st["index_kind"] == 1 and st["n_batches"] == -(-len(reads) // int(batch))
                    assert (np.diff(raw[:, 0].astype(np.int64)) >= 0).all(), "the hit list is not read-major"
                    assert_same(sorted_hits(raw), exp)
    # one batch on a grid of three workgroups: every wave walks some forty wave-tiles in a row, each
    # with a candidate list that spills past LDS -- the lists of the wave-tile in hand and of the one
    # before it are alive together and must not share spill space
    monkeypatch.delenv("MUSC_BATCH_READS", raising=False)
    monkeypatch.delenv("MUSC_MATCH", raising=False)
    monkeypatch.setenv("MUSC_DEBUG_GRID", "3")
    with Engine(0) as eng:
        eng.load_targets(targets)
        eng.load_reads(reads)
        for mode, exp in ((False, exp_all), (True, exp_best)):
            for rep in range(2):
                raw = eng.match(to_cfg(ocfg), apply_mmtol=mode)
                assert eng.stats()["index_kind"] == 1 and eng.stats()["n_batches"] == 1
                assert_same(sorted_hits(raw), exp)


def _x_in_reads_only(seed, n_targets, n_reads, xrate, heavy):
    """synthetic_medium with X (the reference's letter for N) in the reads alone: `xrate` per base,
    and `heavy` reads with five to nine X each."""
    reads, targets = synthetic_medium(seed, n_targets, n_reads)
    rng = np.random.default_rng(seed + 1)
    R = np.array([np.frombuffer(r, dtype=np.uint8) for r in reads]).copy()
    R[rng.random(R.shape) < xrate] = ord("X")
    for i in rng.choice(len(R), size=heavy, replace=False):
        R[i, rng.choice(R.shape[1], size=int(rng.integers(5, 10)), replace=False)] = ord("X")
    return sorted({bytes(r) for r in R}), targets


@pytest.mark.parametrize("pmatch,ww,windows,fits", [(0.97, 15, (0, 20), True), (0.96, 12, (0, 20), True), (0.95, 15, (3, 20), False),
                                                    (0.97, 17, (0, 20), True)])
def test_reads_with_x_against_an_x_free_database(eng, pmatch, ww, windows, fits):
    """Reads with X (N in the FASTQ) against a database without any: on context buckets k_match_t
    treats an X as a mismatch wherever the read lands and keeps windows that hold one from probing
    (kernels_match_lane.hpp, rdx).  A read with more X than its xpos word lists takes part only
    if that many mismatches exceed its budget: 100 bp at PMatch 0.97 / 0.96 allows 3 / 4 < 5..9, at
    0.95 it allows 5 and the run takes the two-kernel path.  Tuples against the literal oracle."""
    from muscato_amd import sorted_hits
    reads, targets = _x_in_reads_only(500 + ww, 3000, 40000, 0.004, 300)
    c = orc.Config(Windows=list(windows), WindowWidth=ww, PMatch=pmatch, MinDinuc=3, MaxReadLength=100,
                   MaxMatches=1000000, MMTol=1)
    rbuf, roff = literal.concat(reads)
    gbuf, goff = literal.concat(targets)
    exp, _, _ = literal.match_arrays(rbuf, roff, gbuf, goff,
                                     literal.make_params(c, bloom_size=128_000_000, num_hash=8, nthreads=8))
    got = gpu_hits(eng, c, reads, targets, False)
    st = eng.stats()
    assert st["index_kind"] == (1 if fits and eng.index_mode in ("auto", "dma") else 3 if eng.index_mode == "lines" else 0)
    assert len(got) > 8000
    assert_same(got, exp)
    best = sorted_hits(eng.match(to_cfg(c), apply_mmtol=True))
    assert_same(best, as_arr(orc.best_filter([tuple(int(x) for x in r) for r in exp], c.MMTol)))


@pytest.mark.parametrize("seed", list(range(0, 60, 3)))
def test_random_cases_with_x_in_the_reads_only(eng, seed):
    """The X-alphabet cases of make_case with the targets' X replaced: ragged lengths, pos-0, target
    ends, one to three windows -- reads with X on whichever path the library picks."""
    ocfg, reads, targets = make_case(seed)
    targets = [t.replace(b"X", b"A") for t in targets]
    full = orc.match_direct(reads, targets, ocfg)
    assert_same(gpu_hits(eng, ocfg, reads, targets, False), as_arr(full))
    from muscato_amd import sorted_hits
    got = sorted_hits(eng.match(to_cfg(ocfg), apply_mmtol=True))
    assert_same(got, as_arr(orc.best_filter(full, ocfg.MMTol)))
