"""Seeded random test cases shared by the oracle cross-check and the GPU parity tests."""
import random

from oracle import muscato_oracle as orc


def rand_seq(rng, n, alphabet):
    return bytes(rng.choice(alphabet) for _ in range(n))


def mutate(rng, s, rate, alphabet):
    b = bytearray(s)
    for i in range(len(b)):
        if rng.random() < rate:
            b[i] = rng.choice(alphabet)
    return bytes(b)


def make_case(seed):
    """Small randomized (cfg, unique sorted reads, targets).

    Alphabet with X for every third seed; 1-3 windows (often starting at 0);
    ragged read lengths incl. reads shorter than a window; reads sampled from
    targets at position 0, at the target end and in between, with 0-8 %
    substitutions; 30 % of targets are mutated copies of earlier ones
    (multi-map).  Every window is guaranteed at least one long-enough read (the
    reference exits otherwise, cmd/muscato_window_reads/main.go:143-151).
    """
    rng = random.Random(seed)
    alphabet = b"ACGT" if seed % 3 else b"ACGTX"
    ww = rng.choice([4, 5, 8, 12, 15])
    nwin = rng.choice([1, 2, 3])
    windows = sorted(rng.sample(range(0, 40), nwin))
    if rng.random() < 0.6:
        windows[0] = 0
    maxlen = rng.choice([60, 100, 120])
    cfg = orc.Config(Windows=windows, WindowWidth=ww,
                     PMatch=rng.choice([1.0, 0.97, 0.93, 0.9, 0.8, 0.7]),
                     MinDinuc=rng.choice([0, 1, 3, 5]), MaxReadLength=maxlen,
                     MaxMatches=1000000, MMTol=rng.choice([0, 1, 3]),
                     MatchMode=rng.choice(["best", "first"]))
    ngene = rng.randint(3, 25)
    targets = []
    for g in range(ngene):
        if targets and rng.random() < 0.3:
            targets.append(mutate(rng, rng.choice(targets), 0.03, alphabet))
        else:
            targets.append(rand_seq(rng, rng.randint(10, 260), alphabet))
    reads = set()
    for _ in range(rng.randint(5, 80)):
        t = rng.choice(targets)
        L = rng.randint(max(1, ww - 2), maxlen)
        u = rng.random()
        if u < 0.15 or len(t) < L:
            r = rand_seq(rng, L, alphabet)
        else:
            p = 0 if u < 0.35 else rng.randint(0, len(t) - L)
            if u > 0.9:
                p = len(t) - L  # flush with the target end
            r = mutate(rng, t[p:p + L], rng.choice([0, 0.01, 0.03, 0.08]), alphabet)
        reads.add(r)
    reads.add(rand_seq(rng, maxlen, alphabet))  # every window has a valid read
    return cfg, sorted(reads), targets
