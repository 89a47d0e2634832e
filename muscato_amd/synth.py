"""Synthetic workloads of SURVEY.md 8(d) / BASELINE.json, generated with torch on the device
that will run them (torch is plumbing here: device memory + RNG, nothing else).

Targets: iid uniform ACGT; ``copy_frac`` of them are copies of a random earlier target with
``copy_sub`` iid substitutions (non-trivial multi-map).
Raw reads, R in total: 70 % sampled from a uniformly random target at a uniform offset
(offset 0 forced for 0.1 % -> the pos-0 path) with 1 %/base iid substitutions; 20 % iid random
(non-matching); 10 % exact duplicates of an earlier read.  The reference de-duplicates before
its hot path (cmd/muscato_uniqify), so the duplicates are dropped by construction: the hot
path sees U = R - R/10 unique reads and reads/sec is quoted on R.

The Go reference's generator (cmd/muscato_gendat/main.go:99-136, unseeded math/rand) cannot be
reproduced bit for bit; ``gendat_like`` mirrors its structure (random reads; gene i < G/2
carries read i%10 at offset i%10) for a tests/bigtest/test.sh-shaped smoke run.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

SEED_BASE = 0x4D555343  # "MUSC"


@dataclass
class Workload:
    name: str
    n_targets: int
    target_len: int
    n_raw_reads: int
    read_len: int
    windows: tuple
    window_width: int
    pmatch: float
    mmtol: int
    min_dinuc: int
    max_matches: int = 1000 * 1000
    match_mode: str = "best"
    revcomp: bool = False  # odd targets are the reverse complements of the even ones (prep_targets -rev)
    total_raw_reads: int = None  # set when the workload fixes the reads over ALL ranks (strong scaling)
    seed_key: str = ""           # what the generator seeds are derived from (the WORKLOADS key)
    gendat: bool = False         # cmd/muscato_gendat's data instead of the generator above (gendat_like)

    @property
    def n_unique_reads(self) -> int:
        return self.n_raw_reads - self.n_raw_reads // 10


# BASELINE.json "configs" (SURVEY.md 8d): cfg2 = configs[1], cfg3 = configs[2], ...
WORKLOADS = {
    "cfg2": Workload("cfg2: 1M reads x 100k targets", 100_000, 1000, 1_000_000, 100, (0, 20), 15, 0.97, 0, 5),
    "cfg3": Workload("cfg3: 50M reads x 1M targets", 1_000_000, 1000, 50_000_000, 100, (0, 20), 15, 0.97, 0, 5),
    # per-GPU shard of cfg4 (200M reads / 8 GPUs) against the same replicated 1M-target DB
    "cfg4shard": Workload("cfg4 shard: 25M reads x 1M targets", 1_000_000, 1000, 25_000_000, 100, (0, 20), 15,
                          0.97, 0, 5),
    # runs beyond 120 bases of context (wide context buckets): cfg3's database with configs[4]'s three
    # windows, and with 150-bp reads
    "cfg3w3": Workload("cfg3 with Windows 0,20,40: 50M reads x 1M targets", 1_000_000, 1000, 50_000_000, 100, (0, 20, 40), 15,
                       0.97, 0, 5),
    "cfg3r150": Workload("cfg3 with 150-bp reads: 25M reads x 1M targets", 1_000_000, 1000, 25_000_000, 150, (0, 20), 15,
                         0.97, 0, 5),
    # the reference's own scale run (tests/bigtest/test.sh:8-14): muscato_gendat -NumRead=100000 -NumGene=100000, then
    # -WindowWidth=20 -Windows=10,30,50,70 with PMatch 1, MinDinuc 0, MMTol 0 (the defaults of cmd/muscato/main.go:855-891)
    "bigtest": Workload("bigtest (tests/bigtest/test.sh): 100k gendat reads x 100k gendat genes", 100_000, 1000, 100_000, 100,
                        (10, 30, 50, 70), 20, 1.0, 0, 0, gendat=True),
    "tiny": Workload("tiny: 20k reads x 2k targets", 2_000, 1000, 20_000, 100, (0, 20), 15, 0.97, 0, 5),
    # per-GPU shard of cfg5: 200M reads / 8 GPUs against 5M targets + reverse complements (10 Gbp),
    # three windows, MMTol=3 ("exhaustive multi-map")
    "cfg5shard": Workload("cfg5 shard: 25M reads x 5M targets with -rev (10M sequences)", 10_000_000, 1000,
                          25_000_000, 100, (0, 20, 40), 15, 0.97, 3, 5, revcomp=True),
}

for _k, _w in WORKLOADS.items():
    _w.seed_key = _k


def workload_for(key: str, world: int = 1) -> Workload:
    """WORKLOADS[key], or the multi-GPU BASELINE configurations whose reads are fixed in total:
    "cfg4" = BASELINE configs[3], 200 M raw reads x 1 M targets sharded over `world` ranks
    (25 M per rank at 8), "cfg5" = configs[4], 200 M reads x 5 M targets + reverse complements,
    Windows 0,20,40, MMTol 3.  Each rank generates its own 200 M / world reads."""
    if key in WORKLOADS:
        return WORKLOADS[key]
    total = 200_000_000
    per = total // max(world, 1)
    if key == "cfg4":
        return Workload("cfg4: 200M reads x 1M targets over %d GPU(s), %dM per rank" % (world, per // 1_000_000),
                        1_000_000, 1000, per, 100, (0, 20), 15, 0.97, 0, 5, total_raw_reads=per * world, seed_key="cfg4")
    if key == "cfg5":
        return Workload("cfg5: 200M reads x 5M targets with -rev (10M sequences) over %d GPU(s), %dM per rank"
                        % (world, per // 1_000_000), 10_000_000, 1000, per, 100, (0, 20, 40), 15, 0.97, 3, 5,
                        revcomp=True, total_raw_reads=per * world, seed_key="cfg5")
    raise KeyError("unknown workload %r (have %s, cfg4, cfg5)" % (key, ", ".join(WORKLOADS)))


_ASCII = (65, 67, 71, 84)  # A C G T


def _lut(device):
    return torch.tensor(_ASCII, dtype=torch.uint8, device=device)


def gen_targets(wl: Workload, device, seed: int, copy_frac: float = 0.2, copy_sub: float = 0.02,
                chunk: int = 100_000) -> torch.Tensor:
    """-> uint8 ASCII tensor [n_targets, target_len] on `device`."""
    if wl.gendat:  # (reads and genes come from one generator: the first ten reads are planted in the genes)
        return gendat_like(wl.n_unique_reads, wl.n_targets, wl.read_len, wl.target_len, device, SEED_BASE + 404)[1]
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    T, L = wl.n_targets, wl.target_len
    out = torch.empty((T, L), dtype=torch.uint8, device=device)
    for s in range(0, T, chunk):
        e = min(T, s + chunk)
        out[s:e] = torch.randint(0, 4, (e - s, L), dtype=torch.uint8, device=device, generator=g)
    ncopy = int(T * copy_frac)
    if ncopy and T - ncopy > 0:
        for s in range(T - ncopy, T, chunk):
            e = min(T, s + chunk)
            src = torch.randint(0, T - ncopy, (e - s,), device=device, generator=g)
            blk = out[src]
            sub = torch.rand((e - s, L), device=device, generator=g) < copy_sub
            rnd = torch.randint(0, 4, (e - s, L), dtype=torch.uint8, device=device, generator=g)
            out[s:e] = torch.where(sub, rnd, blk)
    if wl.revcomp:
        # cmd/muscato_prep_targets/main.go:48-66, 113-131: sequence 2i+1 = reverse complement of 2i
        # (codes A0 C1 G2 T3: complement = 3 - code)
        for s in range(0, T - 1, 2 * chunk):
            e = min(T - 1, s + 2 * chunk)
            ev = out[s:e:2]
            out[s + 1:e + 1:2] = (3 - ev).flip(1)
    lut = _lut(device)
    for s in range(0, T, chunk):
        e = min(T, s + chunk)
        out[s:e] = lut[out[s:e].long()]
    return out


def gen_unique_reads(wl: Workload, targets_ascii: torch.Tensor, device, seed: int, n_unique: int = None,
                     sub_rate: float = 0.01, chunk: int = 1_000_000, return_plan: bool = False):
    """-> uint8 ASCII tensor [U, read_len]: the unique reads the hot path processes
    (7/9 sampled from targets, 2/9 random), in random order.  With return_plan also the
    bookkeeping of where each read came from: (reads, {"gene", "off": int64 [U], "mm": uint8 [U]})
    -- mm = mismatches of the read against its source placement, 255 for the random reads."""
    if wl.gendat and not return_plan:
        return gendat_like(wl.n_unique_reads, wl.n_targets, wl.read_len, wl.target_len, device, SEED_BASE + 404)[0]
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    U = wl.n_unique_reads if n_unique is None else n_unique
    L, TL, T = wl.read_len, wl.target_len, wl.n_targets
    out = torch.empty((U, L), dtype=torch.uint8, device=device)
    plan = None
    if return_plan:
        plan = {"gene": torch.empty(U, dtype=torch.int64, device=device),
                "off": torch.empty(U, dtype=torch.int64, device=device),
                "mm": torch.empty(U, dtype=torch.uint8, device=device)}
    lut = _lut(device)
    ar = torch.arange(L, device=device)
    flat = targets_ascii.reshape(-1)
    for s in range(0, U, chunk):
        e = min(U, s + chunk)
        n = e - s
        gi = torch.randint(0, T, (n,), device=device, generator=g)
        off = torch.randint(0, TL - L + 1, (n,), device=device, generator=g)
        off = torch.where(torch.rand((n,), device=device, generator=g) < 0.001, torch.zeros_like(off), off)
        base = gi * TL + off
        blk = flat[(base[:, None] + ar[None, :])]
        rnd = lut[torch.randint(0, 4, (n, L), device=device, generator=g).long()]
        sub = torch.rand((n, L), device=device, generator=g) < sub_rate
        israndom = torch.rand((n,), device=device, generator=g) < (2.0 / 9.0)
        out[s:e] = torch.where(sub | israndom[:, None], rnd, blk)
        if plan is not None:
            plan["gene"][s:e] = gi
            plan["off"][s:e] = off
            mm = (out[s:e] != blk).sum(dim=1).to(torch.uint8)
            plan["mm"][s:e] = torch.where(israndom, torch.full_like(mm, 255), mm)
    return (out, plan) if return_plan else out


def sort_reads(reads: torch.Tensor) -> torch.Tensor:
    """Bytewise (LC_ALL=C) sort of fixed-length ASCII reads -- the order of the reference's
    reads_sorted.txt.sz, which is what its hot path consumes (cmd/muscato/main.go:180-189).
    LSD radix over 25-base digits; A<C<G<T<X in ASCII, so the codes 0..4 keep the order."""
    n, L = reads.shape
    dev = reads.device
    code = torch.full((256,), 4, dtype=torch.int64, device=dev)  # anything else is the reference's X
    for i, c in enumerate(_ASCII):
        code[c] = i
    perm = torch.arange(n, device=dev)
    nd = (L + 24) // 25
    for d in range(nd - 1, -1, -1):
        lo, hi = d * 25, min(L, d * 25 + 25)
        key = torch.zeros(n, dtype=torch.int64, device=dev)
        for j in range(lo, hi):
            key = key * 5 + code[reads[:, j].long()]
        key = key * (5 ** (25 - (hi - lo)))
        order = torch.sort(key[perm], stable=True).indices
        perm = perm[order]
        del key, order
    return reads[perm]


def offsets_for(n: int, length: int, device) -> torch.Tensor:
    """uint64 offsets [n+1] of fixed-length sequences, stored in an int64 tensor."""
    return torch.arange(0, n + 1, dtype=torch.int64, device=device) * length


def gendat_like(n_reads: int, n_genes: int, read_len: int, gene_len: int, device, seed: int):
    """cmd/muscato_gendat/main.go:99-136 shape: random reads; gene i < G/2 carries read i%10
    at offset i%10."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    lut = _lut(device)
    reads = lut[torch.randint(0, 4, (n_reads, read_len), device=device, generator=g).long()]
    genes = lut[torch.randint(0, 4, (n_genes, gene_len), device=device, generator=g).long()]
    h = n_genes // 2
    for j in range(min(10, h)):
        genes[j:h:10, j:j + read_len] = reads[j]
    return reads, genes
