"""The reference's own scale run as a parity test: tests/bigtest/test.sh:8-14 -- `muscato_gendat -NumRead=100000
-NumGene=100000` (cmd/muscato_gendat/main.go:99-136: random 100-base reads; gene i of the first half carries read
i % 10 at offset i % 10, so ten reads sit in 5 000 genes each), then `muscato -WindowWidth=20
-Windows=10,30,50,70 -MaxReadLength=200` with everything else at its default (PMatch 1, MinDinuc 0, MMTol 0,
MaxMatches 10^6, MatchMode best: cmd/muscato/main.go:855-891).  This is the shape this design is least at home in:
four windows (wide context buckets with two inline entries per line, the four-window k_match_t instance), a hashed
20-mer table, and key blocks of 5 000 entries whose probes walk thousands of overflow entries.

Every accepted tuple of the GPU path against oracle/literal.cpp (the CPU port: Bloom + rolling-hash scan, bytewise
sort, merge-join, cdiff), on the four paths -- what the library picks, MUSC_MATCH=dma (k_match_g is not built for
four windows: k_match_t again, on purpose), MUSC_INDEX=classic and MUSC_INDEX=lines -- at PMatch 1 (the reference's
run) and 0.97."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_READS, N_GENES, READ_LEN, GENE_LEN = 100_000, 100_000, 100, 1000


@pytest.fixture(scope="module")
def gendat():
    import torch
    from muscato_amd import synth
    dev = torch.device("cuda", 0)
    reads, genes = synth.gendat_like(N_READS, N_GENES, READ_LEN, GENE_LEN, dev, synth.SEED_BASE + 404)  # (bench.py --workload bigtest runs the same generator)
    planted = reads[:10].cpu().numpy()
    R = synth.sort_reads(reads)
    keep = torch.ones(R.shape[0], dtype=torch.bool, device=dev)
    keep[1:] = (R[1:] != R[:-1]).any(dim=1)
    R = R[keep].cpu().numpy()
    G = genes.cpu().numpy()
    del reads, genes
    return {"R": R, "G": G, "planted": planted}


@pytest.fixture(scope="module")
def expected(gendat):
    """the CPU port's tuples per PMatch (seconds: 10^8 target positions x 4 windows)"""
    from oracle import literal
    R, G = gendat["R"], gendat["G"]
    rbuf = np.concatenate([R.reshape(-1), np.zeros(8, np.uint8)])
    gbuf = np.concatenate([G.reshape(-1), np.zeros(8, np.uint8)])
    roff = np.arange(len(R) + 1, dtype=np.uint64) * np.uint64(READ_LEN)
    goff = np.arange(len(G) + 1, dtype=np.uint64) * np.uint64(GENE_LEN)
    out = {}
    for pm in (1.0, 0.97):
        class OC:
            Windows = [10, 30, 50, 70]; WindowWidth = 20; PMatch = pm; MinDinuc = 0
            MaxReadLength = 200; MaxMatches = 1_000_000; MatchMode = "best"
        hits, _, _ = literal.match_arrays(rbuf, roff, gbuf, goff,
                                          literal.make_params(OC, bloom_size=1_000_000_000, num_hash=20, nthreads=min(16, os.cpu_count() or 1)))
        out[pm] = hits
    return {"rbuf": rbuf, "roff": roff, "gbuf": gbuf, "goff": goff, "hits": out}


def _keys(a):
    a = a.astype(np.uint64)
    return np.sort((a[:, 0] << np.uint64(40)) | (a[:, 1] << np.uint64(16)) | (a[:, 2] << np.uint64(4)) | a[:, 3])


@pytest.mark.parametrize("mode", ["auto", "dma", "classic", "lines"])
def test_the_reference_bigtest_geometry(mode, gendat, expected, monkeypatch):
    from muscato_amd import Config, Engine
    monkeypatch.delenv("MUSC_INDEX", raising=False)
    monkeypatch.delenv("MUSC_MATCH", raising=False)
    if mode in ("classic", "lines"):
        monkeypatch.setenv("MUSC_INDEX", mode)
    elif mode == "dma":
        monkeypatch.setenv("MUSC_MATCH", "dma")
    # the ten planted reads, among the sorted distinct reads
    R = gendat["R"]
    view = np.ascontiguousarray(R).view([("s", "S%d" % READ_LEN)]).ravel()
    pview = np.ascontiguousarray(gendat["planted"]).view([("s", "S%d" % READ_LEN)]).ravel()
    planted_idx = set(int(np.searchsorted(view, p)) for p in pview)
    with Engine(0) as eng:
        eng.load_targets_arrays(expected["gbuf"], expected["goff"])
        eng.load_reads_arrays(expected["rbuf"], expected["roff"])
        for pm in (1.0, 0.97):
            cfg = Config(Windows=[10, 30, 50, 70], WindowWidth=20, PMatch=pm, MinDinuc=0, MaxReadLength=200,
                         MaxMatches=1_000_000, MMTol=0, MatchMode="best")
            got = eng.match(cfg, apply_mmtol=False)
            st = eng.stats()
            exp = expected["hits"][pm]
            print("[bigtest] %s PMatch %.2f: index kind %d, kernel variant %d, %d tuples (port: %d), %d candidates, %d overflow entries, device %.2f ms"
                  % (mode, pm, st["index_kind"], st["match_variant"], len(got), len(exp), st["n_candidates"], st["n_overflow_entries"], st["ms_total"]), flush=True)
            # four windows spanning 60 + 100 = 160 bases: wide context buckets (the four-window k_match_t instance) when
            # the library picks; the forced two-kernel paths keep theirs
            assert st["index_kind"] == {"auto": 2, "dma": 2, "classic": 0, "lines": 3}[mode]
            if mode in ("auto", "dma"):
                assert st["match_variant"] == 2 and st["n_overflow_entries"] > 100_000
            assert st["n_overflow_blocks"] == 0
            # what the generator planted: read j of the first ten sits at offset j of every tenth gene of the first half
            assert len(exp) >= 10 * (N_GENES // 2 // 10) and set(np.unique(exp[:, 0]).tolist()) >= planted_idx
            assert got.shape == exp.shape, (mode, pm, got.shape, exp.shape)
            assert (_keys(got) == _keys(exp)).all(), (mode, pm)
