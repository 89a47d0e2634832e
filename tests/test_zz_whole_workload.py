"""BASELINE.json configs[2] (cfg3: 50 M raw / 44.8 M distinct reads x 1 M targets) WHOLE, tuple for tuple,
against the CPU port of the reference's algorithm (oracle/literal.cpp: NumHash = 20 rolling hashes, one
4e9-bit Bloom filter per window as cmd/muscato_screen/main.go:116-366 builds and scans them, bytewise candidate
sort, block merge-join and byte-wise cdiff as cmd/muscato_confirm/main.go:151-250, 375-416): all 45.85 M accepted
tuples, not a sample.  About 150 s of CPU on the GPU box's 16 host threads.

The file's name sorts it behind every other test file, and this is its only test: a time limit on the suite can cut
this comparison short but cannot hide any other test.  profiles/cpu_full.py does the same from
the command line (and for the other workloads) and prints the port's stage times."""
import os
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _log(msg):
    print("[whole] " + msg, flush=True)


def test_cfg3_whole_workload_against_the_cpu_port(monkeypatch):
    import torch
    from muscato_amd import Config, Engine, synth
    from oracle import literal

    monkeypatch.delenv("MUSC_MATCH", raising=False)
    monkeypatch.delenv("MUSC_INDEX", raising=False)
    wl = synth.workload_for("cfg3", 1)
    dev = torch.device("cuda", 0)
    seed = synth.SEED_BASE + sum(ord(c) for c in wl.seed_key)
    t0 = time.time()
    T = synth.gen_targets(wl, dev, seed)
    R = synth.sort_reads(synth.gen_unique_reads(wl, T, dev, seed + 7919))
    keep = torch.ones(R.shape[0], dtype=torch.bool, device=dev)
    keep[1:] = (R[1:] != R[:-1]).any(dim=1)
    R = R[keep]
    U, L = R.shape
    NT, TL = T.shape
    rbuf = np.concatenate([R.reshape(-1).cpu().numpy(), np.zeros(8, np.uint8)])
    gbuf = np.concatenate([T.reshape(-1).cpu().numpy(), np.zeros(8, np.uint8)])
    del R, T, keep
    torch.cuda.empty_cache()
    roff = np.arange(U + 1, dtype=np.uint64) * np.uint64(L)
    goff = np.arange(NT + 1, dtype=np.uint64) * np.uint64(TL)
    _log("cfg3: %d distinct reads x %d targets generated in %.0f s" % (U, NT, time.time() - t0))

    # the GPU path first (seconds): every accepted tuple, no best + MMTol selection, on the kernel the library picks
    cfg = Config(Windows=list(wl.windows), WindowWidth=wl.window_width, PMatch=wl.pmatch, MinDinuc=wl.min_dinuc,
                 MaxReadLength=wl.read_len, MaxMatches=wl.max_matches, MMTol=wl.mmtol, MatchMode=wl.match_mode)
    with Engine(0) as eng:
        eng.load_targets_arrays(gbuf, goff)
        eng.load_reads_arrays(rbuf, roff)
        got = eng.match(cfg, apply_mmtol=False)
        st = eng.stats()
    _log("GPU: %d accepted tuples, index kind %d, kernel variant %d, device %.1f ms" % (len(got), st["index_kind"], st["match_variant"], st["ms_total"]))
    assert st["index_kind"] == 1 and st["n_overflow_blocks"] == 0
    # BASELINE's geometry on a direct table: the geometry-specialised k_match_t instance is the one that ran
    assert st["match_variant"] == 3

    class OC:
        Windows = list(wl.windows); WindowWidth = wl.window_width; PMatch = wl.pmatch
        MinDinuc = wl.min_dinuc; MaxReadLength = wl.read_len; MaxMatches = wl.max_matches
        MatchMode = wl.match_mode
    nthr = max(1, min(16, os.cpu_count() or 1))
    done = threading.Event()

    def heartbeat():  # (a sign of life once a minute: a command that writes nothing for seven minutes counts as hung)
        t_start = time.time()
        while not done.wait(60.0):
            _log("... the CPU port has been running for %.0f s" % (time.time() - t_start))
    threading.Thread(target=heartbeat, daemon=True).start()
    t0 = time.time()
    hits, tim, cnt = literal.match_arrays(rbuf, roff, gbuf, goff,
                                          literal.make_params(OC, bloom_size=4_000_000_000, num_hash=20, nthreads=nthr))
    done.set()
    _log("CPU port: %d accepted tuples in %.0f s on %d threads (windows %.0f, Bloom %.0f, scan %.0f, candidate sort %.0f, confirm %.0f s)"
         % ((len(hits), time.time() - t0, nthr) + tuple(float(x) for x in tim)))

    def keys(a):  # (read, gene, pos, nmiss) -> one sortable u64 (26 + 24 + 10 + 4 bits)
        a = a.astype(np.uint64)
        return np.sort((a[:, 0] << np.uint64(38)) | (a[:, 1] << np.uint64(14)) | (a[:, 2] << np.uint64(4)) | a[:, 3])
    assert got.shape == hits.shape, (got.shape, hits.shape)
    assert (keys(got) == keys(hits)).all()
