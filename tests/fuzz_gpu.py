"""Long-running fuzz of the GPU path (not collected by pytest): `python tests/fuzz_gpu.py LO HI`
runs make_case(seed) for seed in [LO, HI) through one Engine -- all accepted tuples, best+MMTol
three times (the later passes are sync-free; with MUSC_GRAPH=1 the last is a graph replay), and the same reads through the GPU read prep -- against
the Python oracle.  Round 1: seeds 0..100000 with the final kernels, no mismatch (270 s on one MI355X); round 2 (context
buckets wherever a case fits them): seeds 0..90000 with k_match and again with k_match_d (up to two
windows; k_match otherwise), no mismatch (308 s each); 30000 of them with MUSC_GRAPH=1; round 3 (k_match_t): seeds 0..60000, no mismatch (153 s)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cases import make_case
from oracle import muscato_oracle as orc
from muscato_amd import Engine, Config, sorted_hits
lo, hi = int(sys.argv[1]), int(sys.argv[2])
e = Engine(0)
bad = 0
t0 = time.time()
for seed in range(lo, hi):
    ocfg, reads, targets = make_case(seed)
    full = sorted(orc.match_direct(reads, targets, ocfg))
    best = sorted(orc.best_filter(full, ocfg.MMTol))
    e.load_targets(targets); e.load_reads(reads)
    cfg = Config(Windows=list(ocfg.Windows), WindowWidth=ocfg.WindowWidth, PMatch=ocfg.PMatch, MinDinuc=ocfg.MinDinuc, MaxReadLength=ocfg.MaxReadLength, MaxMatches=ocfg.MaxMatches, MMTol=ocfg.MMTol, MatchMode=ocfg.MatchMode)
    for mode, exp in ((False, full), (True, best), (True, best), (True, best)):  # sizing, sized, (MUSC_GRAPH=1: captured, replayed)
        got = [tuple(int(x) for x in r) for r in sorted_hits(e.match(cfg, apply_mmtol=mode))]
        if got != exp:
            bad += 1
            print("MISMATCH seed", seed, "apply_mmtol", mode, len(got), len(exp), flush=True)
    # the same reads through the GPU prep, shuffled with duplicates
    raw = list(reads) + list(reads[::3])
    rng = np.random.default_rng(seed); rng.shuffle(raw)
    order, ustart = e.sort_unique_reads(raw)
    if [raw[i] for i in order[ustart[:-1]]] != list(reads):
        bad += 1; print("PREP MISMATCH seed", seed, flush=True)
    got = [tuple(int(x) for x in r) for r in sorted_hits(e.match(cfg, apply_mmtol=False))]
    if got != full:
        bad += 1; print("MISMATCH after prep, seed", seed, flush=True)
    if seed % 200 == 0: print("seed", seed, "elapsed %.0fs" % (time.time() - t0), flush=True)
print("fuzz", lo, hi, "bad", bad, "in %.0fs" % (time.time() - t0))
