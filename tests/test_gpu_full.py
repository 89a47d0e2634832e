"""BASELINE.json configs[2] (cfg3: 50 M reads x 1 M targets) and the per-GPU shard of configs[4]
(cfg5: 25 M reads x 5 M targets + reverse complements = 10 M sequences / 10 Gbp, Windows 0,20,40,
MMTol 3) at their own scale on one MI355X.

(a) size-independent properties of the full run -- every tuple is a real placement whose nmiss
    is the true Hamming distance, the union over windows is a set, planted reads are found at
    their source, no (window,key) block can have overflowed MaxMatches, best + MMTol keeps exactly
    the tuples within MMTol of each read's best (cmd/muscato_confirm/main.go:171-250,
    cmd/muscato_combine_windows/main.go:36-60; `-rev` numbering of
    cmd/muscato_prep_targets/main.go:48-66 for cfg5);
(b) a 900 k-read x 100 k-target sample of the same workload, tuple for tuple, against the literal
    CPU restatement (oracle/literal.cpp).

The reads go through the library's own read prep (musc_reads_sort_unique): the hot path sees the
distinct sequences in bytewise order, as it does behind reads_sorted.txt.sz."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SAMPLE_READS, SAMPLE_TARGETS = 900_000, 100_000


def _log(msg):
    print("[full] " + msg, flush=True)


def _hamming_ok(T_flat, TL, R, h, L, chunk=2_000_000):
    """every tuple in range, and nmiss == Hamming(read, target[pos:pos+L]) -- on the device"""
    import torch
    ar = torch.arange(L, device=h.device)
    for s in range(0, h.shape[0], chunk):
        b = h[s:s + chunk]
        if not bool((b[:, 2] + L <= TL).all()):
            return False
        tsub = T_flat[(b[:, 1] * TL + b[:, 2])[:, None] + ar[None, :]]
        if not bool(((tsub != R[b[:, 0]]).sum(dim=1) == b[:, 3]).all()):
            return False
    return True


@pytest.mark.parametrize("name,index", [("cfg3", "auto"), ("cfg3", "dma"), ("cfg3", "classic"), ("cfg5shard", "auto")])
def test_baseline_config_at_full_size(name, index, monkeypatch):
    """index "auto": what the library picks -- context buckets + the fused k_match for cfg3 (two
    windows, 100-bp reads: 120 bases of context), line buckets + k_screen_t -> k_confirm
    for the cfg5 shard (10 Gbp: positions beyond 32 bits, and the context table would not fit);
    "classic" forces the two-kernel path (on 64-byte buckets: a sparse database) for cfg3 as well."""
    import torch
    monkeypatch.delenv("MUSC_MATCH", raising=False)
    if index == "dma":  # context buckets with k_match_g (LDS-DMA, three to four waves per SIMD) instead of k_match_t
        monkeypatch.setenv("MUSC_MATCH", "dma")
        monkeypatch.delenv("MUSC_INDEX", raising=False)
    elif index == "classic":
        monkeypatch.setenv("MUSC_INDEX", "classic")
    else:
        monkeypatch.delenv("MUSC_INDEX", raising=False)
    want_kind = 1 if name == "cfg3" and index in ("auto", "dma") else 3 if name == "cfg5shard" else 0  # cfg5: line buckets
    from muscato_amd import Config, Engine, sorted_hits, synth
    from oracle import literal
    from oracle import muscato_oracle as orc

    wl = synth.WORKLOADS[name]
    dev = torch.device("cuda", 0)
    L, TL, NT = wl.read_len, wl.target_len, wl.n_targets
    budget = int((1.0 - wl.pmatch) * L)
    seed = synth.SEED_BASE + sum(ord(c) for c in name)
    cfg = Config(Windows=list(wl.windows), WindowWidth=wl.window_width, PMatch=wl.pmatch, MinDinuc=wl.min_dinuc,
                 MaxReadLength=L, MaxMatches=wl.max_matches, MMTol=wl.mmtol, MatchMode=wl.match_mode)

    t0 = time.time()
    T = synth.gen_targets(wl, dev, seed)
    toff = synth.offsets_for(NT, TL, dev)
    raw, plan = synth.gen_unique_reads(wl, T, dev, seed + 7919, return_plan=True)
    nraw = raw.shape[0]
    roff = synth.offsets_for(nraw, L, dev)
    torch.cuda.synchronize()
    _log("%s: %d targets, %d generated reads in %.1fs" % (name, NT, nraw, time.time() - t0))

    eng = Engine(0)
    try:
        eng.load_targets_device(T.data_ptr(), toff.data_ptr(), NT)
        eng.build_index_for(cfg, L)
        order, ustart = eng.sort_unique_reads_arrays(raw.data_ptr(), roff.data_ptr(), nraw, True)
        U = len(ustart) - 1
        assert eng.n_reads == U and nraw - U < 0.02 * nraw
        heads = torch.from_numpy(order[ustart[:-1]].astype(np.int64)).to(dev)
        del order, ustart
        R = raw[heads]                          # the distinct reads in the order the library holds them
        del raw, roff
        gene, off, mm = plan["gene"][heads], plan["off"][heads], plan["mm"][heads]
        del plan, heads
        # the prep sorted them bytewise and collapsed repeats: strictly increasing rows
        for s in range(0, U - 1, 4_000_000):
            a, b = R[s:s + 4_000_000], R[s + 1:s + 4_000_001]
            a = a[:b.shape[0]]
            d = a != b
            first = d.to(torch.uint8).argmax(dim=1, keepdim=True)
            assert bool(d.any(dim=1).all()) and bool((a.gather(1, first) < b.gather(1, first)).all())
        del a, b, d, first

        # ---- (a) the full run, every accepted tuple (no MMTol)
        t0 = time.time()
        n_all = eng.match_device(cfg, apply_mmtol=False)
        st = eng.stats()
        _log("%s (index kind %d, %.1f GiB): %d reads -> %d candidates, %d pairs, %d accepted tuples; device %.2f ms "
             "(first pass, %.1fs wall)" % (name, st["index_kind"], st["index_bytes"] / 2**30, U, st["n_candidates"],
                                            st["n_pairs"], n_all, st["ms_total"], time.time() - t0))
        assert st["n_reads"] == U and st["n_overflow_blocks"] == 0 and st["index_kind"] == want_kind
        assert st["n_candidates"] >= st["n_pairs"] >= st["n_accepted"] == n_all > 0.5 * U
        h32 = torch.empty((n_all, 4), dtype=torch.int32, device=dev)
        eng.hits_to(h32.data_ptr(), n_all, True)
        h = h32.to(torch.int64)
        del h32
        assert bool((h[:, 0] < U).all()) and bool((h[:, 1] < NT).all()) and bool((h[:, 3] <= budget).all())
        assert _hamming_ok(T.reshape(-1), TL, R, h, L), "a tuple is not a real placement / nmiss is not the Hamming distance"
        key = (h[:, 0] << 34) | (h[:, 1] << 10) | h[:, 2]   # 26 + 24 + 10 bits (gene < 2^24, pos < 2^10)
        assert NT < (1 << 24) and TL <= (1 << 10) and U < (1 << 26)
        skey = torch.sort(key).values
        assert bool((skey[1:] != skey[:-1]).all()), "the union over windows is not a set"
        # the same pass again: sized now (no host round trips; with k_match_t the tuples of a batch
        # are moved into place by the next batch's launch) -- the same list, tuple for tuple
        assert eng.match_device(cfg, apply_mmtol=False) == n_all
        st2 = eng.stats()
        assert st2["n_batches"] == st["n_batches"] and st2["n_pairs"] == st["n_pairs"]
        h2 = torch.empty((n_all, 4), dtype=torch.int32, device=dev)
        eng.hits_to(h2.data_ptr(), n_all, True)
        assert bool((h2[:, 0].to(torch.int64)[1:] >= h2[:, 0].to(torch.int64)[:-1]).all()), "not read-major"
        h2 = h2.to(torch.int64)
        skey2 = torch.sort((h2[:, 0] << 34) | (h2[:, 1] << 10) | h2[:, 2]).values
        assert bool((skey2 == skey).all()), "the sized pass returns different tuples"
        del h2, skey2
        # planted reads: one that matches its source verbatim (not at target position 0, where the
        # literal-100 rule blinds window 0 for reads longer than 100 - ww) must be reported there
        # with nmiss 0 unless none of its windows passes the MinDinuc gate
        must = (mm == 0) & (off != 0)
        want = (torch.arange(U, device=dev) << 34) | (gene << 10) | off
        pos_in = torch.searchsorted(skey, want).clamp(max=skey.shape[0] - 1)
        found = skey[pos_in] == want
        missing = torch.nonzero(must & ~found).reshape(-1)
        assert int(missing.shape[0]) < 0.01 * int(must.sum())
        ocfg = orc.Config(Windows=list(wl.windows), WindowWidth=wl.window_width, MinDinuc=wl.min_dinuc, MaxReadLength=L)
        for m in missing[:200].tolist():
            r = bytes(R[m].cpu().numpy())
            assert not any(orc.window_valid(r, k, ocfg) for k in range(len(wl.windows))), \
                "a planted read with a usable window was not found"
        # a planted read within budget is reported at its source with its true distance whenever found
        within = (mm <= budget) & found
        assert int(within.sum()) > 0.6 * U
        del want, pos_in, found, must, missing, within

        # best + MMTol: exactly the tuples within MMTol of the read's best
        bmin = torch.full((U,), 255, dtype=torch.int64, device=dev)
        bmin.scatter_reduce_(0, h[:, 0], h[:, 3], reduce="amin")
        keep = h[:, 3] <= bmin[h[:, 0]] + wl.mmtol
        exp_keys = torch.sort(key[keep]).values
        del keep, bmin, key, skey, h
        n_best = eng.match_device(cfg, apply_mmtol=True)
        st2 = eng.stats()
        assert st2["n_overflow_blocks"] == 0
        b32 = torch.empty((n_best, 4), dtype=torch.int32, device=dev)
        eng.hits_to(b32.data_ptr(), n_best, True)
        b = b32.to(torch.int64)
        got_keys = torch.sort((b[:, 0] << 34) | (b[:, 1] << 10) | b[:, 2]).values
        assert got_keys.shape == exp_keys.shape and bool((got_keys == exp_keys).all()), "best+MMTol set differs"
        del b32, b, got_keys, exp_keys
        # idempotence (this pass is "sized": no host round trips)
        assert eng.match_device(cfg, apply_mmtol=True) == n_best
        _log("%s: %d tuples after best+MMTol(%d); sized pass %.2f ms on the device"
             % (name, n_best, wl.mmtol, eng.stats()["ms_total"]))

        # ---- (b) the sample against the literal oracle: the reads planted in the first
        # SAMPLE_TARGETS targets (so that most of the sample has something to find) topped up with
        # evenly spaced others, in the library's order (still distinct and bytewise sorted)
        nt = min(NT, SAMPLE_TARGETS)
        planted = torch.nonzero((gene < nt) & (mm != 255)).reshape(-1)[:2 * SAMPLE_READS // 3]
        stride = max(1, U // (SAMPLE_READS - planted.shape[0]))
        pick = torch.zeros(U, dtype=torch.bool, device=dev)
        pick[planted] = True
        pick[::stride] = True
        idx = torch.nonzero(pick).reshape(-1)[:SAMPLE_READS]
        Rs = R[idx].contiguous()
        ns = Rs.shape[0]
        rbuf = np.concatenate([Rs.reshape(-1).cpu().numpy(), np.zeros(8, np.uint8)])
        gbuf = np.concatenate([T[:nt].reshape(-1).cpu().numpy(), np.zeros(8, np.uint8)])
        rso = np.arange(ns + 1, dtype=np.uint64) * np.uint64(L)
        gso = np.arange(nt + 1, dtype=np.uint64) * np.uint64(TL)
        del planted, pick, idx
    finally:
        eng.close()
    del R, T, Rs, gene, off, mm
    torch.cuda.empty_cache()

    class OC:
        Windows = list(wl.windows); WindowWidth = wl.window_width; PMatch = wl.pmatch
        MinDinuc = wl.min_dinuc; MaxReadLength = L; MaxMatches = wl.max_matches; MatchMode = wl.match_mode
    t0 = time.time()
    exp, _, _ = literal.match_arrays(rbuf, rso, gbuf, gso,
                                     literal.make_params(OC, bloom_size=400_000_000, num_hash=20, nthreads=16))
    _log("%s: literal oracle on %d reads x %d targets: %d tuples in %.1fs" % (name, ns, nt, len(exp), time.time() - t0))
    # a sanity bound on the SAMPLE's usefulness, not on parity: it must hold enough tuples to mean
    # something.  0.1 per sampled read: cfg3's sample has ~0.8 tuples per read, the cfg5 shard's ~0.19
    # (the sample's targets are 1 % of the shard's 10 M sequences, so most planted reads lose theirs)
    assert len(exp) > 0.1 * ns
    with Engine(0) as e2:
        e2.load_targets_arrays(gbuf, gso)
        e2.load_reads_arrays(rbuf, rso)
        got = sorted_hits(e2.match(cfg, apply_mmtol=False))
        assert e2.stats()["n_overflow_blocks"] == 0
        assert got.shape == exp.shape and (got == exp).all(), "GPU tuples differ from the literal oracle on the sample"
        bestg = sorted_hits(e2.match(cfg, apply_mmtol=True))
    bmin = np.full(ns, 255, dtype=np.int64)
    np.minimum.at(bmin, exp[:, 0], exp[:, 3].astype(np.int64))
    bexp = exp[exp[:, 3] <= bmin[exp[:, 0]] + wl.mmtol]
    assert bestg.shape == bexp.shape and (bestg == bexp).all(), "best+MMTol differs from the oracle on the sample"


def test_cfg3_with_x_in_the_database_at_full_size(monkeypatch):
    """BASELINE configs[2] with 0.1 % of the database's bases X (N in the FASTA), at full size: the
    context-bucket path (index kind 1: k_match_t<.., XM = 2>, flagged entries) and the two-kernel path
    (MUSC_NO_X_CONTEXT=1: mask planes in k_confirm) are two implementations of cdiff's X rule
    (cmd/muscato_confirm/main.go:151-159) -- they must return the same tuples, every one a real
    placement whose nmiss is the Hamming distance with every X of the target counted as a mismatch
    (the reads hold none), the union a set, no MaxMatches overflow; and a sample agrees with the
    literal oracle tuple for tuple."""
    import torch
    from muscato_amd import Config, Engine, sorted_hits, synth
    from oracle import literal
    monkeypatch.delenv("MUSC_MATCH", raising=False)
    monkeypatch.delenv("MUSC_INDEX", raising=False)
    monkeypatch.delenv("MUSC_NO_X_CONTEXT", raising=False)
    wl = synth.WORKLOADS["cfg3"]
    dev = torch.device("cuda", 0)
    L, TL, NT = wl.read_len, wl.target_len, wl.n_targets
    budget = int((1.0 - wl.pmatch) * L)
    seed = synth.SEED_BASE + 4242
    cfg = Config(Windows=list(wl.windows), WindowWidth=wl.window_width, PMatch=wl.pmatch, MinDinuc=wl.min_dinuc,
                 MaxReadLength=L, MaxMatches=wl.max_matches, MMTol=wl.mmtol, MatchMode=wl.match_mode)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    T = synth.gen_targets(wl, dev, seed)
    for s0 in range(0, NT, 100_000):
        blk = T[s0:s0 + 100_000]
        blk[torch.rand(blk.shape, device=dev, generator=g) < 0.001] = ord("X")
    # a few runs of X as well (the edge of a run gives contexts with three and more X: the mask-plane path)
    starts = torch.randint(0, NT * TL - 64, (20_000,), device=dev, generator=g)
    flat = T.reshape(-1)
    for d in range(40):
        flat[starts + d] = ord("X")
    toff = synth.offsets_for(NT, TL, dev)
    raw = synth.gen_unique_reads(wl, T, dev, seed + 7919)
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    for s0 in range(0, raw.shape[0], 1_000_000):
        blk = raw[s0:s0 + 1_000_000]
        isx = blk == ord("X")
        blk[isx] = acgt[torch.randint(0, 4, (int(isx.sum()),), device=dev, generator=g)]
    nraw = raw.shape[0]
    roff = synth.offsets_for(nraw, L, dev)
    keys = {}
    eng = Engine(0)
    try:
        eng.load_targets_device(T.data_ptr(), toff.data_ptr(), NT)
        order, ustart = eng.sort_unique_reads_arrays(raw.data_ptr(), roff.data_ptr(), nraw, True)
        U = len(ustart) - 1
        R = raw[torch.from_numpy(order[ustart[:-1]].astype(np.int64)).to(dev)]
        del raw, roff, order, ustart
        for path in ("context", "two-kernel"):
            if path == "two-kernel":
                monkeypatch.setenv("MUSC_NO_X_CONTEXT", "1")
                eng.reload_env()  # (the library reads its MUSC_* knobs once per context)
            n_all = eng.match_device(cfg, apply_mmtol=False)
            st = eng.stats()
            _log("cfg3 + X in the database, %s path (index kind %d): %d reads -> %d accepted tuples, device %.2f ms"
                 % (path, st["index_kind"], U, n_all, st["ms_total"]))
            assert st["index_kind"] == (1 if path == "context" else 0) and st["n_overflow_blocks"] == 0 and n_all > 0.5 * U
            h32 = torch.empty((n_all, 4), dtype=torch.int32, device=dev)
            eng.hits_to(h32.data_ptr(), n_all, True)
            h = h32.to(torch.int64)
            del h32
            assert bool((h[:, 0] < U).all()) and bool((h[:, 1] < NT).all()) and bool((h[:, 3] <= budget).all())
            if path == "context":
                assert _hamming_ok(T.reshape(-1), TL, R, h, L), "nmiss is not the Hamming distance (X of the target = mismatch)"
                # the X matter: some accepted placements cover one
                ar = torch.arange(L, device=dev)
                sub = h[:2_000_000]
                covers = (T.reshape(-1)[(sub[:, 1] * TL + sub[:, 2])[:, None] + ar[None, :]] == ord("X")).any(dim=1)
                assert int(covers.sum()) > 1000
                del sub, covers
            k = torch.sort((h[:, 0] << 36) | (h[:, 1] << 12) | (h[:, 2] << 2) | h[:, 3]).values
            assert bool((k[1:] >> 2 != k[:-1] >> 2).all()), "the union over windows is not a set"
            keys[path] = k
            del h
        assert keys["context"].shape == keys["two-kernel"].shape and bool((keys["context"] == keys["two-kernel"]).all()), \
            "the two paths disagree on a database with X"
        keys.clear()
        # a sample against the literal oracle (context path)
        monkeypatch.delenv("MUSC_NO_X_CONTEXT")
        nt, ns = 50_000, 400_000
        Rs = R[:: max(1, U // ns)][:ns].contiguous()
        ns = Rs.shape[0]
        rbuf = np.concatenate([Rs.reshape(-1).cpu().numpy(), np.zeros(8, np.uint8)])
        gbuf = np.concatenate([T[:nt].reshape(-1).cpu().numpy(), np.zeros(8, np.uint8)])
        rso = np.arange(ns + 1, dtype=np.uint64) * np.uint64(L)
        gso = np.arange(nt + 1, dtype=np.uint64) * np.uint64(TL)
    finally:
        eng.close()
    del R, T, Rs
    torch.cuda.empty_cache()

    class OC:
        Windows = list(wl.windows); WindowWidth = wl.window_width; PMatch = wl.pmatch
        MinDinuc = wl.min_dinuc; MaxReadLength = L; MaxMatches = wl.max_matches; MatchMode = wl.match_mode
    exp, _, _ = literal.match_arrays(rbuf, rso, gbuf, gso, literal.make_params(OC, bloom_size=200_000_000, num_hash=20, nthreads=16))
    assert len(exp) > 0.02 * ns
    with Engine(0) as e2:
        e2.load_targets_arrays(gbuf, gso)
        e2.load_reads_arrays(rbuf, rso)
        got = sorted_hits(e2.match(cfg, apply_mmtol=False))
        assert e2.stats()["index_kind"] == 1 and e2.stats()["n_overflow_blocks"] == 0
        assert got.shape == exp.shape and (got == exp).all(), "GPU tuples differ from the literal oracle on the sample"
    _log("cfg3 + X in the database: sample of %d reads x %d targets: %d tuples equal the literal oracle's" % (ns, nt, len(exp)))
