"""Medium-scale fuzz of the GPU path against the literal C++ oracle (not collected by pytest):
`python tests/fuzz_gpu_medium.py LO HI` draws one random configuration per seed -- 1-5 windows,
WindowWidth 6-20 (direct and hashed index), PMatch 0.9-1, MinDinuc 0-6, MMTol 0-3, X rate 0 / 0.1 % /
1 %, read length 40-150 -- over a few thousand reads and a few hundred targets, and compares all
accepted tuples and the best+MMTol selection.  Round 1: seeds 0..85000, no mismatch (28 min on
one MI355X); round 2 (both index kinds, as each configuration selects): seeds 0..56000 with k_match,
0..80000 with k_match_d where a configuration has at most two windows, and 0..20000 with
MUSC_FUZZ_READS_X=1 (X in the reads only): no mismatch; round 3 (k_match_t, wide and line buckets): seeds
0..22000 and 0..9000 with MUSC_FUZZ_READS_X=1, no mismatch, all four index kinds used; round 4: profiles/r04_fuzz_totals.txt (k_match_t with the
eight-lanes-per-line fetch, and k_match_g with MUSC_MATCH=dma)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import literal
from oracle import muscato_oracle as orc
from muscato_amd import Config, Engine, sorted_hits


# MUSC_FUZZ_READS_X=1: X in the reads only (an X-free database: with at most two windows and reads
# of at most 112 bases those runs take k_match_d's RX path whenever every read's X fit its xpos word)
READS_X_ONLY = bool(os.environ.get("MUSC_FUZZ_READS_X"))
# MUSC_FUZZ_DB_X=1: X in the database only, single bases and runs (N stretches), the reads sampled over
# them get random bases there -- those runs stay on context buckets (k_match_t<.., XM = 2>: flagged
# entries compared through the mask plane); =2: the reads keep the target's X where it falls outside
# their windows (at most three per read), so read X meets target X
DB_X = int(os.environ.get("MUSC_FUZZ_DB_X", "0"))
KINDS = {0: 0, 1: 0, 2: 0, 3: 0}
VARIANTS = {0: 0, 2: 0, 3: 0, 4: 0, 5: 0}  # musc_stats.match_variant: 0 two-kernel path, 2 / 3 k_match_t general / specialised, 4 / 5 k_match_g


def case(seed):
    rng = np.random.default_rng(seed)
    L = int(rng.choice([40, 60, 100, 120, 150]))
    ww = int(rng.integers(6, 21))
    nwin = int(rng.integers(1, 6))
    wins = sorted(int(x) for x in rng.choice(np.arange(0, max(1, L - ww - 5)), size=nwin, replace=False))
    if rng.random() < 0.5:
        wins[0] = 0
    cfg = orc.Config(Windows=wins, WindowWidth=ww, PMatch=float(rng.choice([1.0, 0.97, 0.95, 0.92, 0.9])),
                     MinDinuc=int(rng.integers(0, 7)), MaxReadLength=L, MaxMatches=1000000,
                     MMTol=int(rng.integers(0, 4)), MatchMode=str(rng.choice(["best", "first"])))
    nt, tlen, nr = int(rng.integers(50, 400)), int(rng.integers(L + 5, 800)), int(rng.integers(500, 6000))
    xrate = float(rng.choice([0.0, 0.001, 0.01]))
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    T = bases[rng.integers(0, 4, size=(nt, tlen))]
    ncopy = nt // 4
    T[nt - ncopy:] = T[rng.integers(0, nt - ncopy, size=ncopy)]
    sub = rng.random((ncopy, tlen)) < 0.03
    T[nt - ncopy:][sub] = bases[rng.integers(0, 4, size=int(sub.sum()))]
    if DB_X:
        xrate = float(rng.choice([0.001, 0.004, 0.015]))
        run = int(rng.choice([1, 1, 4, 40]))
        starts = rng.random(T.shape) < xrate / run
        for d in range(run):
            T[:, d:][starts[:, :tlen - d]] = ord("X")
    elif xrate and not READS_X_ONLY:
        T[rng.random(T.shape) < xrate] = ord("X")
    g = rng.integers(0, nt, size=nr)
    p = rng.integers(0, tlen - L + 1, size=nr)
    p[rng.random(nr) < 0.05] = 0
    p[rng.random(nr) < 0.05] = tlen - L
    R = T[g[:, None], p[:, None] + np.arange(L)[None, :]].copy()
    sub = rng.random(R.shape) < float(rng.choice([0.0, 0.01, 0.03]))
    R[sub] = bases[rng.integers(0, 4, size=int(sub.sum()))]
    if DB_X:
        isx = R == ord("X")
        keep = np.zeros_like(isx)
        if DB_X == 2:  # a read keeps the target's X outside its windows, three at most
            inwin = np.zeros(L, dtype=bool)
            for q in wins:
                inwin[q:q + ww] = True
            keep = isx & ~inwin[None, :]
            keep &= np.cumsum(keep, axis=1) <= 3
        fill = isx & ~keep
        R[fill] = bases[rng.integers(0, 4, size=int(fill.sum()))]
        xrate = 0.0
    if READS_X_ONLY:  # X (N in the FASTQ) in the reads alone, at rates where most reads keep a few of them
        xrate = float(rng.choice([0.002, 0.005, 0.02]))
    if xrate:
        R[rng.random(R.shape) < xrate] = ord("X")
    lens = rng.integers(max(ww, L // 2), L + 1, size=nr)
    reads = sorted({bytes(r[:n]) for r, n in zip(R, lens)})
    return cfg, reads, [bytes(t) for t in T]


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    e = Engine(0)
    bad, t0 = 0, time.time()
    for seed in range(lo, hi):
        c, reads, targets = case(seed)
        rbuf, roff = literal.concat(reads)
        gbuf, goff = literal.concat(targets)
        exp, _, _ = literal.match_arrays(rbuf, roff, gbuf, goff, literal.make_params(c, bloom_size=16_000_000, num_hash=6, nthreads=8))
        e.load_targets(targets)
        e.load_reads(reads)
        k = Config(Windows=list(c.Windows), WindowWidth=c.WindowWidth, PMatch=c.PMatch, MinDinuc=c.MinDinuc,
                   MaxReadLength=c.MaxReadLength, MaxMatches=c.MaxMatches, MMTol=c.MMTol, MatchMode=c.MatchMode)
        got = sorted_hits(e.match(k, apply_mmtol=False))
        ok = got.shape == exp.shape and bool((got == exp).all())
        best = sorted_hits(e.match(k, apply_mmtol=True))
        eb = np.array(sorted(orc.best_filter([tuple(int(x) for x in r) for r in exp], c.MMTol)), dtype=np.uint32).reshape(-1, 4)
        ok = ok and best.shape == eb.shape and bool((best == eb).all()) and e.stats()["n_overflow_blocks"] == 0
        KINDS[e.stats()["index_kind"]] += 1
        VARIANTS[e.stats()["match_variant"]] = VARIANTS.get(e.stats()["match_variant"], 0) + 1
        if not ok:
            bad += 1
            print("MISMATCH seed", seed, c, len(reads), len(targets), len(got), len(exp), flush=True)
        if seed % 50 == 0:
            print("seed", seed, "hits", len(exp), "elapsed %.0fs" % (time.time() - t0), flush=True)
    print("fuzz_medium", lo, hi, "bad", bad, "in %.0fs" % (time.time() - t0), "index kinds used", KINDS, "kernel variants", VARIANTS,
          "(reads-only X)" if READS_X_ONLY else "(database X, mode %d)" % DB_X if DB_X else "")


if __name__ == "__main__":
    main()
