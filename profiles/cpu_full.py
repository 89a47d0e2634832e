#!/usr/bin/env python3
"""The CPU port of the reference's algorithm (oracle/literal.cpp: NumHash=20 rolling hashes, one
4e9-bit Bloom filter per window, full target scan, bytewise candidate sort, block merge-join,
byte-wise cdiff) timed on a WHOLE bench workload, not a sample ("configs 2 and 3
in full").  Run once per round through gpurun (the synthetic data is generated on the GPU, as
bench.py does); the line goes to profiles/r<round>_cpu_full.json and bench.py quotes the latest beside its
bounded sample.

    python3 profiles/cpu_full.py cfg3 [threads] [xdb | xreads]     (X in the database / in the reads -> profiles/r<round>_cpu_full_xdb.json / _xreads.json)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch
    from muscato_amd import synth
    from oracle import literal

    key = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    nthr = int(sys.argv[2]) if len(sys.argv) > 2 else max(1, min(16, os.cpu_count() or 1))
    wl = synth.workload_for(key, 1)
    dev = torch.device("cuda", 0)
    seed = synth.SEED_BASE + sum(ord(c) for c in wl.seed_key)
    T = synth.gen_targets(wl, dev, seed)
    xdb = len(sys.argv) > 3 and sys.argv[3] == "xdb"
    if xdb:  # 0.1 % of the database's bases X one by one, plus 20 000 runs of 40 X (N stretches)
        g = torch.Generator(device=dev)
        g.manual_seed(seed + 99)
        for s0 in range(0, wl.n_targets, 100_000):
            blk = T[s0:s0 + 100_000]
            blk[torch.rand(blk.shape, device=dev, generator=g) < 0.001] = ord("X")
        starts = torch.randint(0, T.numel() - 64, (20_000,), device=dev, generator=g)
        flat = T.reshape(-1)
        for d in range(40):
            flat[starts + d] = ord("X")
    R = synth.gen_unique_reads(wl, T, dev, seed + 7919)
    if xdb:  # the reads sampled over an X get a random base there
        acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
        for s0 in range(0, R.shape[0], 1_000_000):
            blk = R[s0:s0 + 1_000_000]
            isx = blk == ord("X")
            blk[isx] = acgt[torch.randint(0, 4, (int(isx.sum()),), device=dev, generator=g)]
    xreads = len(sys.argv) > 3 and sys.argv[3] == "xreads"
    if xreads:  # 0.1 % of the read bases X (N in the FASTQ), the database without
        g = torch.Generator(device=dev)
        g.manual_seed(seed + 199)
        for s0 in range(0, R.shape[0], 1_000_000):
            blk = R[s0:s0 + 1_000_000]
            blk[torch.rand(blk.shape, device=dev, generator=g) < 0.001] = ord("X")
    R = synth.sort_reads(R)
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

    class OC:
        Windows = list(wl.windows); WindowWidth = wl.window_width; PMatch = wl.pmatch
        MinDinuc = wl.min_dinuc; MaxReadLength = wl.read_len; MaxMatches = wl.max_matches
        MatchMode = wl.match_mode
    print("[cpu_full] %s: %d distinct reads x %d targets on %d threads ..." % (wl.name, U, NT, nthr), file=sys.stderr, flush=True)
    # (a sign of life once a minute: the port runs for minutes without a word, and the GPU box's
    # supervisor takes a command that writes nothing for seven minutes to be hung)
    import threading
    done = threading.Event()

    def heartbeat():
        t_start = time.time()
        while not done.wait(60.0):
            print("[cpu_full] ... the port has been running for %.0f s" % (time.time() - t_start), file=sys.stderr, flush=True)
    hb = threading.Thread(target=heartbeat, daemon=True)
    hb.start()
    t0 = time.time()
    hits, tim, cnt = literal.match_arrays(rbuf, roff, gbuf, goff,
                                          literal.make_params(OC, bloom_size=4_000_000_000, num_hash=20, nthreads=nthr))
    wall = time.time() - t0
    done.set()
    t_win, t_bloom, t_scan, t_csort, t_conf = [float(x) for x in tim]
    print("[cpu_full] %.1fs; %d tuples; now the GPU path on the same arrays ..." % (wall, len(hits)), file=sys.stderr, flush=True)
    # the whole workload through the GPU path, tuple for tuple against the port (every accepted tuple, no MMTol)
    from muscato_amd import Config, Engine, sorted_hits
    cfg = Config(Windows=list(wl.windows), WindowWidth=wl.window_width, PMatch=wl.pmatch, MinDinuc=wl.min_dinuc,
                 MaxReadLength=wl.read_len, MaxMatches=wl.max_matches, MMTol=wl.mmtol, MatchMode=wl.match_mode)
    with Engine(0) as eng:
        eng.load_targets_arrays(gbuf, goff)
        eng.load_reads_arrays(rbuf, roff)
        got = eng.match(cfg, apply_mmtol=False)
        gpu_ms = eng.stats()["ms_total"]
        index_kind = eng.stats()["index_kind"]
    def keys(a):  # (read, gene, pos, nmiss) -> one sortable u64 (26 + 24 + 10 + 4 bits)
        a = a.astype(np.uint64)
        return np.sort((a[:, 0] << np.uint64(38)) | (a[:, 1] << np.uint64(14)) | (a[:, 2] << np.uint64(4)) | a[:, 3])
    equal = bool(got.shape == hits.shape and (keys(got) == keys(hits)).all())
    print(json.dumps({wl.name + (" + X in the database (0.1 % one by one, 20 000 runs of 40)" if xdb else " + X in the reads (0.1 % of their bases)" if xreads else ""): {
        "gpu_index_kind": index_kind,
        "reads_per_s": wl.n_raw_reads / wall, "wall_s": wall, "cores": nthr, "kind": "port",
        "raw_reads": wl.n_raw_reads, "distinct_reads": int(U), "targets": int(NT),
        "stages_s": {"windows": t_win, "bloom": t_bloom, "scan": t_scan, "candidate_sort": t_csort, "confirm": t_conf},
        "accepted_tuples": int(len(hits)), "gpu_tuples_identical_on_whole_workload": equal, "gpu_first_pass_device_ms": gpu_ms,
        "what": "oracle/literal.cpp on the whole workload, in-memory arrays in and tuples out (no text / snappy / GNU sort I/O)",
    }}), flush=True)


if __name__ == "__main__":
    main()
