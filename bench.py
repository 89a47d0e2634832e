#!/usr/bin/env python3
"""bench.py -- reads/sec of muscato's seed-and-extend hot path (screen + confirm + per-read
best filter) on MI355X, with the kernels' HBM rooflines and a CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|cfg4|cfg5|...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" (`value`, `ms_per_step`; since r04) is one pass on SURVEY.md 8d's timer scope: this rank's packed
unique reads in pinned host memory -> asynchronous upload -> pack + match against the resident target
database + k-mer index -> the tuples in pinned host memory (N>1: every rank's tuples to its own pinned host
memory AND concatenated on rank 0 over RCCL, inside the timed region).  Beside it the line carries
  * `kernel_pipeline` the pass with reads, database and index resident in HBM and the hits left there (N>1:
                    gathered on rank 0): r01-r03's `value`; `roofline` and `per_step` describe this pass,
  * `survey_scope`  the parts of the step (serial run, what the overlap hides, bytes both ways),
  * `graph_replay`  the sized pass replayed as one hipGraph launch (MUSC_GRAPH=1, opt-in),
  * `first_pass_ms` one pass over freshly loaded reads (the sizing pass the CLI always takes),
    `cold_pass_ms`  the first pass of the process (buffer allocation included),
  * `roofline`      the dominant kernel against the HBM roofline, on the bytes it loads,
    `roofline_screen` / `roofline_confirm` when the two-kernel path runs,
  * `cpu_baseline`  the CPU port of the reference's algorithm on a bounded sample.

Workloads (muscato_amd/synth.py): N=1 defaults to cfg3 (BASELINE configs[2], the largest
single-GPU configuration); N>1 defaults to cfg4 (configs[3]: 200 M reads in total over the N
ranks, strong scaling); --workload cfg3 at N>1 gives every rank its own 50 M reads (weak).
The database upload/pack and the one-off index build are timed separately (reported, not in
`value`).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_GBS = 6290.0        # same guide: 6.29 TB/s measured float4 copy
RANDOM_LINES_PER_S = 51.5e9  # measured on this part: random whole 128-byte lines, eight lanes per line (the shape the kernels use since r04; the quad shape of r02 / r03 reached 46.5-47.6 G), nothing else in the kernel (profiles/ub_dma_lines.hip, profiles/r04_ub_dma_lines.txt)


def line_rate(roof):
    """The kernels of this path fetch random 128-byte lines: how many per second the recorded PMC
    traffic of the dominant kernel amounts to, against the part's measured random-line rate."""
    if roof and roof.get("traffic") and roof.get("avg_launch_ms"):
        lps = roof["traffic"] / 128.0 / (roof["avg_launch_ms"] / 1e3)
        roof["lines_per_s_from_traffic"] = lps
        roof["frac_of_measured_random_line_rate"] = lps / RANDOM_LINES_PER_S
    return roof
TRAFFIC_FILE = os.path.join("profiles", "r04_traffic.json")


def measured_traffic(workload_key, kernel):
    """HBM bytes per launch of `kernel` from the PMC passes recorded under profiles/ (rocprofv3
    --pmc cannot run inside this process); None if no record matches the workload."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            return json.load(f)[workload_key][kernel]["traffic_bytes_per_launch"]
    except Exception:
        return None


def cpu_baseline(wl, cfg, targets_ascii, reads_ascii, eng_factory, n_raw_full, log):
    """Reference-shaped CPU port (oracle/literal.cpp: Bloom + rolling-hash screen, sorted
    merge-join confirm) timed on a bounded sample, plus a bit-exactness check of the GPU path on
    that same sample.  This is the ONLY place bench.py touches oracle/."""
    import numpy as np
    from oracle import literal

    T, TL = targets_ascii.shape
    U, L = reads_ascii.shape
    ncores = os.cpu_count() or 1
    nthr = max(1, min(16, ncores))
    tp = min(T, 100_000)            # target prefix scanned by the CPU
    s = min(U, 900_000)             # unique reads of the sample (= 1M raw reads)
    gbuf = targets_ascii[:tp].reshape(-1).cpu().numpy()
    stride = max(1, U // s)          # evenly spaced sample (the reads are sorted)
    rbuf = reads_ascii[::stride][:s].contiguous().reshape(-1).cpu().numpy()
    gbuf = np.concatenate([gbuf, np.zeros(8, np.uint8)])
    rbuf = np.concatenate([rbuf, np.zeros(8, np.uint8)])
    goff = (np.arange(tp + 1, dtype=np.uint64) * np.uint64(TL))
    roff = (np.arange(s + 1, dtype=np.uint64) * np.uint64(L))

    class OC:  # what literal.make_params reads
        Windows = list(wl.windows); WindowWidth = wl.window_width; PMatch = wl.pmatch
        MinDinuc = wl.min_dinuc; MaxReadLength = wl.read_len; MaxMatches = wl.max_matches
        MatchMode = wl.match_mode
    params = literal.make_params(OC, bloom_size=4_000_000_000, num_hash=20, nthreads=nthr)
    t0 = time.time()
    exp, tim, cnt = literal.match_arrays(rbuf, roff, gbuf, goff, params)
    wall = time.time() - t0
    t_win, t_bloom, t_scan, t_csort, t_conf = [float(x) for x in tim]
    fr = U / s                      # reads scale-up to this rank's full batch
    ft = T / tp                     # database scale-up
    t_full = t_scan * ft + (t_win + t_bloom) * fr + (t_csort + t_conf) * fr * ft
    value = n_raw_full / t_full if t_full > 0 else 0.0
    measured = (s + s // 9) / wall if wall > 0 else 0.0   # the sample itself: raw reads / wall

    # the same port on ONE thread (SURVEY.md 8d asks for both), on a smaller sample to stay bounded
    s1, tp1 = min(s, 100_000), min(tp, 20_000)
    st1 = max(1, s // s1)
    rb1 = np.concatenate([rbuf[:s * L].reshape(s, L)[::st1][:s1].reshape(-1), np.zeros(8, np.uint8)])
    gb1 = np.concatenate([gbuf[:tp1 * TL], np.zeros(8, np.uint8)])
    p1 = literal.make_params(OC, bloom_size=4_000_000_000, num_hash=20, nthreads=1)
    t1 = time.time()
    _, tim1, _ = literal.match_arrays(rb1, np.arange(s1 + 1, dtype=np.uint64) * np.uint64(L), gb1,
                                      np.arange(tp1 + 1, dtype=np.uint64) * np.uint64(TL), p1)
    wall1 = time.time() - t1
    w1, b1, sc1, cs1, cf1 = [float(x) for x in tim1]
    t_full1 = sc1 * (T / tp1) + (w1 + b1) * (U / s1) + (cs1 + cf1) * (U / s1) * (T / tp1)
    single = {"value": (s1 + s1 // 9) / wall1 if wall1 > 0 else 0.0, "cores": 1,
              "extrapolated_full_reads_per_s": n_raw_full / t_full1 if t_full1 > 0 else 0.0,
              "sample": "%d unique reads (=%d raw) x %d targets through the same port on one thread, %.1fs wall"
                        % (s1, s1 + s1 // 9, tp1, wall1)}

    # bit-exactness of the GPU path on the very same sample (all accepted tuples, no MMTol)
    from muscato_amd import sorted_hits
    eng = eng_factory()
    eng.load_targets_arrays(gbuf, goff)
    eng.load_reads_arrays(rbuf, roff)
    got = sorted_hits(eng.match(cfg, apply_mmtol=False))
    eng.close()
    exact = bool(got.shape == exp.shape and (got == exp).all())
    log("cpu sample: %d reads x %d targets, %d threads: window %.2fs bloom %.2fs scan %.2fs "
        "candsort %.2fs confirm %.2fs (wall %.2fs); %d hits; gpu bit-exact on sample: %s"
        % (s, tp, nthr, t_win, t_bloom, t_scan, t_csort, t_conf, wall, len(exp), exact))
    full, full_src = None, None
    for cand in ("r04_cpu_full.json", "r03_cpu_full.json", "r02_cpu_full.json"):  # the same port timed on the whole cfg3 workload, once per round, through gpurun (profiles/cpu_full.py)
        try:
            with open(os.path.join(ROOT, "profiles", cand)) as f:
                full = json.load(f).get(wl.name)
            if full:
                full_src = "profiles/" + cand
                break
        except Exception:
            pass
    return {
        # `value` is a TIMING: the raw reads of the sample over the wall time of the port on it (the
        # sample is BASELINE cfg2's size).  The whole workload through the same port is timed once per
        # round (profiles/cpu_full.py) and quoted as `full_workload_measured`; the extrapolation of
        # the sample's stage times to the full batch is kept as `extrapolated_full_reads_per_s` only.
        "value": measured, "unit": "reads/s", "cores": nthr, "kind": "port",
        "scope": "port, kernels only: in-memory arrays in, tuples out; no FASTQ/snappy/text/GNU sort I/O",
        "sample": ("%d evenly spaced unique reads (=%d raw) x first %d of %d targets through oracle/literal.cpp "
                   "(NumHash=20, BloomSize=4e9) on %d threads: %.2fs wall (scan %.2fs, windows+bloom %.2fs, candidate "
                   "sort+confirm %.2fs)" % (s, s + s // 9, tp, T, nthr, wall, t_scan, t_win + t_bloom, t_csort + t_conf)),
        "sample_wall_s": wall, "extrapolated_full_reads_per_s": value,
        "extrapolation": "scan*%.0f + read terms*%.1f + pair terms*%.1f*%.0f = %.1fs" % (ft, fr, fr, ft, t_full),
        "full_workload_measured": full, "full_workload_source": full_src,
        "gpu_bit_exact_on_sample": exact, "sample_hits": int(len(exp)),
        "single_thread": single,
    }


def pack2bit_device(ascii2d):
    """[U, L] ASCII on the device -> the ABI's packed form (2 bits per base A0 C1 G2 T3, 4 bases
    per byte, little-endian within the byte) as a uint8 device tensor.  No X in timing runs."""
    import torch
    dev = ascii2d.device
    lut = torch.zeros(256, dtype=torch.uint8, device=dev)
    for i, c in enumerate(b"ACGT"):
        lut[c] = i
    flat = ascii2d.reshape(-1)
    n = flat.shape[0]
    out = torch.empty((n + 3) // 4 + 8, dtype=torch.uint8, device=dev)
    out[(n + 3) // 4:] = 0
    step = 1 << 28
    for s in range(0, n, step):
        e = min(n, s + step)
        c = lut[flat[s:e].long()]
        if (e - s) % 4:
            c = torch.cat([c, torch.zeros(4 - (e - s) % 4, dtype=torch.uint8, device=dev)])
        c = c.reshape(-1, 4)
        out[s // 4:s // 4 + c.shape[0]] = c[:, 0] | (c[:, 1] << 2) | (c[:, 2] << 4) | (c[:, 3] << 6)
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=None, help="default: cfg3 on one GPU, cfg4 (200 M reads in total) on several")
    ap.add_argument("--reads", type=int, default=0, help="override raw reads per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-survey-scope", action="store_true", help="skip the pinned-host-to-pinned-host leg")
    ap.add_argument("--unsorted", action="store_true", help="leave the unique reads in random order")
    ap.add_argument("--no-block-check", action="store_true", help="skip the MaxMatches per-block overflow check")
    ap.add_argument("--index", choices=["auto", "classic"], default="auto",
                    help="classic: force the 64-byte-bucket index and the two-kernel path (MUSC_INDEX=classic)")
    ap.add_argument("--x-reads-only", action="store_true",
                    help="with --xrate: X (N in the FASTQ) in the reads alone, the database stays X-free (reads with X on context buckets)")
    ap.add_argument("--x-db-only", action="store_true",
                    help="with --xrate: X (N in the FASTA) in the database alone; the reads sampled over one get a random base there")
    ap.add_argument("--xrate", type=float, default=0.0,
                    help="fraction of bases replaced by X in targets and reads (the correctness/timing run with the mask planes)")
    args = ap.parse_args()

    if args.index == "classic":
        os.environ["MUSC_INDEX"] = "classic"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs a torch.distributed.run launch with that many ranks" % args.gpus,
                  file=sys.stderr)
            return 2
        args.gpus = world
    if args.workload is None:
        args.workload = "cfg3" if world == 1 else "cfg4"

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    import torch
    import torch.distributed as dist
    from muscato_amd import Config, Engine, synth

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the hot path has no CPU fallback", file=sys.stderr)
        return 3
    # Rehearsal knobs (never set by the driver): run N ranks on ONE card over gloo to exercise
    # the N>1 control flow where only one GPU exists (RCCL refuses two ranks on one device).
    backend = os.environ.get("MUSC_BENCH_BACKEND", "nccl")
    # MUSC_BENCH_FORCE_DIST=1 with one rank: a one-member RCCL group, so that the N>1 code (device
    # buffers filled by the library, counts agreed by all_reduce on the device, the gatherer's
    # stream handling) runs on a one-GPU box; only the transfers themselves have no peer
    multi = world > 1 or bool(os.environ.get("MUSC_BENCH_FORCE_DIST"))
    if "MUSC_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["MUSC_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    wl = synth.workload_for(args.workload, world)
    strong = wl.total_raw_reads is not None
    if args.reads:
        wl = synth.Workload(**{**wl.__dict__, "n_raw_reads": args.reads, "total_raw_reads": None,
                               "name": wl.name + " (reads=%d)" % args.reads})
        strong = False
    cfg = Config(Windows=list(wl.windows), WindowWidth=wl.window_width, PMatch=wl.pmatch,
                 MinDinuc=wl.min_dinuc, MaxReadLength=wl.read_len, MaxMatches=wl.max_matches,
                 MMTol=wl.mmtol, MatchMode=wl.match_mode)
    seed = synth.SEED_BASE + sum(ord(c) for c in wl.seed_key)

    t0 = time.time()
    targets = synth.gen_targets(wl, device, seed)
    if args.xrate > 0:
        gx = torch.Generator(device=device)
        gx.manual_seed(seed + 99)
        for s0 in range(0, 0 if args.x_reads_only else wl.n_targets, 100_000):
            blk = targets[s0:s0 + 100_000]
            blk[torch.rand(blk.shape, device=device, generator=gx) < args.xrate] = ord("X")
    toff = synth.offsets_for(wl.n_targets, wl.target_len, device)
    torch.cuda.synchronize()
    log("generated %d targets x %d bp in %.1fs" % (wl.n_targets, wl.target_len, time.time() - t0))

    eng = Engine(local_rank)
    t0 = time.time()
    eng.load_targets_device(targets.data_ptr(), toff.data_ptr(), wl.n_targets)
    t_dbload = time.time() - t0
    t0 = time.time()
    eng.build_index_for(cfg, wl.read_len)  # the index the match will pick (context buckets when the run fits them)
    t_index = time.time() - t0
    ms_index = eng.stats()["ms_index_build"]
    log("db pack %.3fs, index build %.3fs (device %.1f ms)" % (t_dbload, t_index, ms_index))

    t0 = time.time()
    U = wl.n_unique_reads
    reads = synth.gen_unique_reads(wl, targets, device, seed + 7919 * (rank + 1))
    if args.xrate > 0 and args.x_db_only:
        acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
        for s0 in range(0, reads.shape[0], 1_000_000):
            blk = reads[s0:s0 + 1_000_000]
            isx = blk == ord("X")
            blk[isx] = acgt[torch.randint(0, 4, (int(isx.sum()),), device=device, generator=gx)]
    elif args.xrate > 0:
        for s0 in range(0, reads.shape[0], 1_000_000):
            blk = reads[s0:s0 + 1_000_000]
            blk[torch.rand(blk.shape, device=device, generator=gx) < args.xrate] = ord("X")
    if not args.unsorted:
        reads = synth.sort_reads(reads)  # the hot path's input is reads_sorted.txt.sz
    roff = synth.offsets_for(U, wl.read_len, device)
    torch.cuda.synchronize()
    log("generated %d unique reads (%d raw) in %.1fs" % (U, wl.n_raw_reads, time.time() - t0))
    # Read prep through the library (untimed, one-off): the raw reads = the unique ones + 10 %
    # duplicates in random order go through musc_reads_sort_unique, which leaves the distinct
    # reads loaded in bytewise order -- the hot path's input -- and is checked against torch's sort.
    prep = {}
    if args.unsorted:
        eng.load_reads_device(reads.data_ptr(), roff.data_ptr(), U)
    else:
        g = torch.Generator(device=device)
        g.manual_seed(seed + 17)
        n_raw = wl.n_raw_reads
        extra = torch.randint(0, U, (n_raw - U,), device=device, generator=g)
        raw = torch.cat([reads, reads[extra]])[torch.randperm(n_raw, device=device, generator=g)]
        del extra
        raw_off = synth.offsets_for(n_raw, wl.read_len, device)
        torch.cuda.synchronize()
        t1 = time.time()
        order, ustart = eng.sort_unique_reads_arrays(raw.data_ptr(), raw_off.data_ptr(), n_raw, True)
        prep = {"read_prep_wall_s": time.time() - t1, "read_prep_device_ms": eng.stats()["ms_read_prep"],
                "raw_reads": n_raw, "distinct": int(len(ustart) - 1)}
        # check: the group heads are strictly increasing rows (sorted, no repeats) and as many as the
        # torch-sorted reads have distinct rows (the generator's "unique" reads repeat now and then)
        heads = torch.from_numpy(order[ustart[:-1]].astype("int64")).to(device)
        A = raw[heads]
        d = A[1:] != A[:-1]
        first = d.to(torch.uint8).argmax(dim=1, keepdim=True)
        increasing = bool(d.any(dim=1).all() and (A[:-1].gather(1, first) < A[1:].gather(1, first)).all())
        del d, first, A
        keep = torch.ones(U, dtype=torch.bool, device=device)
        keep[1:] = (reads[1:] != reads[:-1]).any(dim=1)
        n_distinct = int(keep.sum())
        prep["equals_torch_sort"] = bool(increasing and n_distinct == len(ustart) - 1)
        reads = reads[keep]  # the distinct reads, as the library now holds them
        del raw, raw_off, heads, order, ustart, keep
        log("read prep (sort + collapse) of %d raw reads -> %d distinct: %.1f ms on the device, %.2f s wall incl. "
            "the order download; sorted, distinct and as many as torch's sort finds: %s"
            % (n_raw, prep["distinct"], prep["read_prep_device_ms"], prep["read_prep_wall_s"], prep["equals_torch_sort"]))
    del roff, toff
    n_loaded = eng.n_reads
    keep_for_cpu = (rank == 0 and not multi and not args.no_cpu_baseline)
    scope_on = not args.no_survey_scope and not args.unsorted and args.xrate == 0
    keep_reads = keep_for_cpu or scope_on
    if not keep_for_cpu:
        del targets
    if not keep_reads:
        del reads
    torch.cuda.empty_cache()

    def match():
        return eng.match_device(cfg, apply_mmtol=True, skip_block_check=args.no_block_check, n_shards=world)

    # the first pass of the process: sizing + every buffer allocated on the way
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n0 = match()
    cold_pass_ms = (time.perf_counter() - t1) * 1e3
    cold_device_ms = eng.stats()["ms_total"]

    read_base = rank * n_loaded if not strong else None
    if multi:
        # global read numbers: rank r's reads follow those of ranks 0..r-1
        cnt = torch.tensor([n_loaded], dtype=torch.int64, device=device if backend == "nccl" else "cpu")
        cnts = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(cnts, cnt)
        loaded = [int(c.item()) for c in cnts]
        read_base = sum(loaded[:rank])
        total_loaded = sum(loaded)
    else:
        read_base, total_loaded = 0, n_loaded
    gathered_n = [0]
    gatherer = None
    gather_mode = "none"
    if multi:
        # The tuples of every pass end up on rank 0 in rank order (= global read order).  The
        # gather of pass i (grouped send/recv into rank 0's rank-major buffer, counts ride in the
        # buffers) runs on the communicator's stream while pass i+1 is matched; capacity agreed once.
        from muscato_amd.dist import HitGatherer
        gdev = device if backend == "nccl" else torch.device("cpu")
        # 8-byte tuples on the links when the fields fit 64 bits: global read number, target
        # number, position, mismatch count (rank 0 ingests 7 shards' tuples per pass at 8 GPUs)
        budget = int((1.0 - wl.pmatch) * wl.read_len)
        pack_bits = [max(1, (total_loaded - 1).bit_length()), max(1, (wl.n_targets - 1).bit_length()),
                     max(1, wl.target_len.bit_length()), max(1, budget.bit_length())]
        use_packed = sum(pack_bits) <= 64 and not os.environ.get("MUSC_BENCH_UNPACKED")
        # the compact form when gene | pos | nmiss fit one u32 word: the read index rides as one count
        # byte per read (the list is read-major) -- 5.2 bytes per tuple at cfg4 against 8
        use_compact = sum(pack_bits[1:]) <= 32 and use_packed and not os.environ.get("MUSC_BENCH_NO_COMPACT")
        # every rank's capacity, agreed ONCE (the sizing pass's counts + 5 %): a pass then posts its transfers without a
        # collective or a host read (r03 agreed row counts per pass)
        rank_caps = HitGatherer.agree_caps(n0, gdev)
        cap = max(rank_caps)
        if use_compact:
            # every rank must be able to use the form (no read with more than 255 tuples, fields fit):
            # tried once on the pass that sized the buffers, agreed over all ranks
            ok = 1
            try:
                tw = torch.empty(max(n0, 1), dtype=torch.int32, device=device)
                tc = torch.empty(max(n_loaded, 4), dtype=torch.uint8, device=device)
                eng.hits_to_compact(tw.data_ptr(), max(n0, 1), tc.data_ptr(), max(n_loaded, 4), True, pack_bits[1:])
                del tw, tc
            except Exception as e:
                log("compact tuples not usable on this rank (%r): 8-byte form" % (e,))
                ok = 0
            okt = torch.tensor([ok], dtype=torch.int64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
            use_compact = bool(int(okt.item()))
        if use_compact:
            gatherer = HitGatherer(cap, gdev, compact_reads=max(loaded), rank_caps=rank_caps)
        else:
            gatherer = HitGatherer(cap, gdev, packed=use_packed, rank_caps=rank_caps)
        gather_mode = "overlapped (HitGatherer, grouped send/recv), " + (
            "compact tuples: u32 word %s + one count byte per read" % pack_bits[1:] if use_compact else
            "8-byte packed tuples %s" % pack_bits if use_packed else "16-byte tuples")

    overflow_seen = [0]

    def fill_gather(buf, n):
        if gatherer.compact_reads:
            counts, words = gatherer.compact_views(buf)
            eng.hits_to_compact(words.data_ptr(), gatherer.cap, counts.data_ptr(), gatherer.compact_reads,
                                buf.is_cuda, pack_bits[1:])
            return n, n_loaded
        if n and gatherer.packed:
            eng.hits_to_packed(buf.data_ptr(), n, buf.is_cuda, pack_bits, read_base)
        elif n:
            eng.hits_to(buf.data_ptr(), n, buf.is_cuda)
        return n

    def step():
        n = match()
        overflow_seen[0] = max(overflow_seen[0], eng.stats()["n_overflow_blocks"])
        if gatherer is not None:
            gatherer.submit(lambda buf: fill_gather(buf, n), read_base)  # an error here is fatal on purpose: a rank that fell back
            # to another collective on its own would leave the others waiting in a different one
        else:
            gathered_n[0] = n
        return n

    def barrier():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    acc = {"ms_confirm": 0.0, "launches": 0, "bytes": 0, "ms_screen": 0.0, "ms_scan": 0.0,
           "ms_select": 0.0, "ms_total": 0.0, "screen_launches": 0, "match_launches": 0, "match_bytes": 0,
           "match_bytes_strict": 0}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        st = eng.stats()  # host-side read of numbers the library already holds
        acc["ms_confirm"] += st["ms_confirm"]; acc["launches"] += st["confirm_launches"]
        acc["bytes"] += st["confirm_bytes"]; acc["screen_launches"] += st["n_batches"]
        for k in ("match_launches", "match_bytes", "match_bytes_strict"):
            acc[k] += st[k]
        for k in ("ms_screen", "ms_scan", "ms_select", "ms_total"):
            acc[k] += st[k]
    if gatherer is not None:
        cnts = gatherer.finish()  # the last gathers complete inside the timed region
        if cnts is not None:
            gathered_n[0] = sum(cnts)
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the MaxMatches proof holds for the union of the shards only if it holds on every shard
        o = torch.tensor([min(overflow_seen[0], 2 ** 62)], dtype=torch.int64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(o, op=dist.ReduceOp.MAX)
        overflow_seen[0] = int(o.item())
    st = eng.stats()

    # ---- the timed region `value` is quoted on (r04): SURVEY.md 8d's scope -- per step: this rank's
    # packed unique reads (2 bits per base, fixed length) in PINNED HOST memory -> musc_reads_load_packed32(async):
    # the upload queued in pieces on a copy stream -> musc_match_device packs and matches each batch as its pieces
    # arrive (a sizing pass: the reads are new to the context) -> the tuples on the host: one GPU -- into pinned host
    # memory, compact (u32 word + a count byte per read) where gene | pos | nmiss fit 32 bits, 8-byte packed words
    # where read | gene | pos | nmiss fit 64 (cfg5), else 16-byte tuples; several GPUs -- gathered on rank 0 over RCCL
    # every step (HitGatherer), the last step's slabs copied to rank 0's pinned host memory inside the region.
    # Database + index resident (uploaded once: one_off).  EXACTLY K steps between barrier + synchronize.
    nmax = int(n0 * 1.05) + 16
    scope = None
    h_hits = h_packed = h_words = h_counts = h_w64 = None
    if scope_on:
        packed_dev = pack2bit_device(reads)
        h_packed = torch.empty(packed_dev.shape, dtype=torch.uint8, pin_memory=True)
        h_packed.copy_(packed_dev)
        del packed_dev
        budget = int((1.0 - wl.pmatch) * wl.read_len)
        cbits = [max(1, (wl.n_targets - 1).bit_length()), max(1, wl.target_len.bit_length()), max(1, budget.bit_length())]
        rbits = [max(1, (max(n_loaded, 2) - 1).bit_length())] + cbits
        form = "compact" if sum(cbits) <= 32 else "packed64" if sum(rbits) <= 64 else "plain"
        if not multi:
            if form == "compact":
                h_words = torch.empty(nmax, dtype=torch.int32, pin_memory=True)
                h_counts = torch.empty(n_loaded + 8, dtype=torch.uint8, pin_memory=True)
            elif form == "packed64":
                h_w64 = torch.empty(nmax, dtype=torch.int64, pin_memory=True)
            h_hits = torch.empty((nmax, 4), dtype=torch.int32, pin_memory=True)
        h_slabs = None
        scope_t = {"queue_upload": 0.0, "match": 0.0, "down": 0.0}

        def download(n):
            nonlocal form
            if n and form == "compact":
                try:
                    eng.hits_to_compact(h_words.data_ptr(), nmax, h_counts.data_ptr(), n_loaded + 8, False, cbits)
                    return
                except Exception as e:  # a read with more than 255 tuples: the next form
                    log("compact tuples not usable (%r)" % (e,))
                    form = "plain"
            if n and form == "packed64":
                eng.hits_to_packed(h_w64.data_ptr(), n, False, rbits, 0)
            elif n:
                eng.hits_to(h_hits.data_ptr(), n, False)

        # Several GPUs: SURVEY.md 8d stops the clock at "result tuples for all shards on rank 0's host".  Rank 0 owns ONE
        # pinned host buffer with a slab per rank (a file in /dev/shm that every rank maps and registers with the HIP
        # runtime); every step each rank copies the buffer it has just handed to the gather -- header + count bytes +
        # words, or packed words -- into its slab over ITS OWN PCIe link, while RCCL concatenates the same buffers in rank
        # 0's HBM.  (Rank 0 pulling all ranks' tuples out of its HBM instead would put 8 shards' bytes per step on one link.)
        h_all = [None, None, None]  # the mapping (kept alive), this rank's slab, the file's path

        def open_host_slabs():
            import mmap
            import numpy as np
            esz = gatherer.send[0].element_size() * (4 if gatherer.send[0].dim() == 2 else 1)
            sizes = [((r_ * esz + 4095) // 4096) * 4096 for r_ in gatherer.rows]  # page-aligned slabs
            path = [None]
            if rank == 0:
                path[0] = "/dev/shm/musc_bench_%d_%s" % (os.getpid(), os.environ.get("MASTER_PORT", "0"))
                try:
                    with open(path[0], "wb") as f:
                        os.posix_fallocate(f.fileno(), 0, sum(sizes))  # (reserved now: a mapping of a sparse file dies with SIGBUS when /dev/shm is full)
                except OSError as e:
                    log("no room for the shared host buffer (%r): every rank keeps its tuples in a pinned buffer of its own" % (e,))
                    try:
                        os.unlink(path[0])
                    except OSError:
                        pass
                    path[0] = None
            dist.broadcast_object_list(path, src=0)
            if path[0] is None:
                src0 = gatherer.send[0][:gatherer.rows[rank]]
                h_all[1] = torch.empty(src0.shape, dtype=src0.dtype, pin_memory=src0.is_cuda)
                h_all[0] = (None, None, None, bool(src0.is_cuda))
                return
            with open(path[0], "r+b") as f:
                mm = mmap.mmap(f.fileno(), sum(sizes))
            arr = np.frombuffer(mm, dtype=np.uint8)
            whole = torch.from_numpy(arr)
            pinned = False
            if gatherer.send[0].is_cuda:
                try:
                    pinned = int(torch.cuda.cudart().cudaHostRegister(whole.data_ptr(), whole.numel(), 0)) == 0
                except Exception as e:  # pageable then: slower, still correct
                    log("host slabs not registered (%r): pageable copies" % (e,))
            off = sum(sizes[:rank])
            mine = whole[off:off + gatherer.rows[rank] * esz].view(gatherer.send[0].dtype)
            if gatherer.send[0].dim() == 2:
                mine = mine.view(-1, 4)
            h_all[0], h_all[1], h_all[2] = (mm, arr, whole, pinned), mine, path[0]
            dist.barrier()
            if rank == 0:
                os.unlink(path[0])  # (every rank holds its mapping; the name is no longer needed)

        def own_to_host():
            k_ = (gatherer.i - 1) % gatherer.depth
            h_all[1].copy_(gatherer.send[k_][:gatherer.rows[rank]])

        if gatherer is not None:
            open_host_slabs()

        def scope_step(async_upload=True):
            t1 = time.perf_counter()
            eng.load_reads_packed32_ptr(h_packed.data_ptr(), 0, 0, wl.read_len, n_loaded, async_upload=async_upload)
            t2 = time.perf_counter()
            n = match()
            t3 = time.perf_counter()
            overflow_seen[0] = max(overflow_seen[0], eng.stats()["n_overflow_blocks"])
            if gatherer is not None:
                gatherer.submit(lambda buf: fill_gather(buf, n), read_base)
                own_to_host()
            else:
                download(n)
                gathered_n[0] = n
            t4 = time.perf_counter()
            scope_t["queue_upload"] += t2 - t1; scope_t["match"] += t3 - t2; scope_t["down"] += t4 - t3
            return n

        assert scope_step() == n0, "the scope pass and the resident pass disagree"  # warm-up (buffers of a sizing pass)
        if gatherer is not None:
            gatherer.finish()
        for k_ in scope_t:
            scope_t[k_] = 0.0
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            n = scope_step()
            assert n == n0
        if gatherer is not None:
            cnts = gatherer.finish()
            if cnts is not None:
                gathered_n[0] = sum(cnts)
                slabs = gatherer.last_slabs()
                if h_slabs is None:
                    h_slabs = torch.empty(slabs.shape, dtype=slabs.dtype, pin_memory=slabs.is_cuda)
                h_slabs.copy_(slabs)
        barrier()
        scope_elapsed = time.perf_counter() - t0
        if multi:
            t = torch.tensor([scope_elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            scope_elapsed = float(t.item())
            o = torch.tensor([min(overflow_seen[0], 2 ** 62)], dtype=torch.int64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(o, op=dist.ReduceOp.MAX)
            overflow_seen[0] = int(o.item())
        scope = {"elapsed": scope_elapsed, "ms_per_step": scope_elapsed * 1e3 / args.steps, "form": form, "cbits": cbits, "rbits": rbits,
                 "host_ms_queue_upload": scope_t["queue_upload"] * 1e3 / args.steps, "host_ms_match_call": scope_t["match"] * 1e3 / args.steps,
                 "host_ms_tuples_out": scope_t["down"] * 1e3 / args.steps,
                 "bytes_up": int(h_packed.numel()),
                 "bytes_down": (int(h_all[1].numel() * h_all[1].element_size()) if h_all[1] is not None else
                                int(n0 * 4 + n_loaded) if form == "compact" else int(n0 * 8) if form == "packed64" else int(n0 * 16)),
                 "host_slabs_pinned": (bool(h_all[0][3]) if h_all[0] is not None else None)}
        match()  # (the legs below start from a sized pass again)

    legs = {}
    if not multi:
        import numpy as np
        # (1) steady-state pass + D2H of the tuples into PINNED host memory (the r01 figure used pageable)
        if h_hits is None:
            h_hits = torch.empty((nmax, 4), dtype=torch.int32, pin_memory=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n = match()
        if n:
            eng.hits_to(h_hits.data_ptr(), n, False)
        legs["pcie_inclusive"] = {"ms_per_step": (time.perf_counter() - t1) * 1e3,
                                  "what": "sized pass + tuples copied to pinned host memory (no read upload)"}
        legs["pcie_inclusive"]["reads_per_s"] = wl.n_raw_reads / (legs["pcie_inclusive"]["ms_per_step"] / 1e3)

        if st["index_kind"] in (1, 2):
            # (1b) the same sized pass replayed as one hipGraph launch (MUSC_GRAPH=1, opt-in): what is
            # left of the host's share once the seven launches per batch are one
            os.environ["MUSC_GRAPH"] = "1"
            eng.reload_env()  # (the knobs are read once per context; the next pass sizes itself again)
            try:
                for _ in range(3):  # sizing pass, capture, first replay
                    assert match() == n0
                torch.cuda.synchronize()
                dev_ms = 0.0
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    assert match() == n0
                    dev_ms += eng.stats()["ms_total"]
                wall = (time.perf_counter() - t1) * 1e3 / args.steps
                legs["graph_replay"] = {"ms_per_pass": wall, "device_ms": dev_ms / args.steps,
                                        "host_overhead_ms": wall - dev_ms / args.steps,
                                        "reads_per_s": wl.n_raw_reads / (wall / 1e3),
                                        "what": "MUSC_GRAPH=1: sized pass as one hipGraphLaunch + one stream sync"}
            finally:
                os.environ.pop("MUSC_GRAPH", None)
                eng.reload_env()
                match()  # (sized again for the legs below)

        if scope is not None:
            # (2) the same three steps of SURVEY.md 8d's scope one after the other (the upload finished before the match
            # starts, the download after it): what the pipelined form of the timed region hides.  (A copy + kernel trace
            # would show the same on a time line; rocprofv3 --memory-copy-trace dies at exit on this image, profiles/README.md.)
            serial = []
            for rep in range(2):
                t1 = time.perf_counter()
                eng.load_reads_packed32_ptr(h_packed.data_ptr(), 0, 0, wl.read_len, n_loaded, async_upload=False)
                t_up = time.perf_counter()
                n = match()
                t_m = time.perf_counter()
                download(n)
                t_dn = time.perf_counter()
                serial.append(((t_dn - t1) * 1e3, (t_up - t1) * 1e3, (t_m - t_up) * 1e3, (t_dn - t_m) * 1e3))
                assert n == n0
            sbest = min(serial)
            legs["survey_scope"] = {
                "serial": {"ms_per_pass": sbest[0], "ms_upload_then_pack": sbest[1], "ms_match": sbest[2], "ms_tuples_to_host": sbest[3],
                           "what": "the same steps without overlap (blocking upload, then the match, then the download); best of 2"},
                "ms_hidden_by_overlap": sbest[0] - scope["ms_per_step"],
                "ms_per_pass": scope["ms_per_step"], "reads_per_s": wl.n_raw_reads / (scope["ms_per_step"] / 1e3),
                "host_ms_queue_upload": scope["host_ms_queue_upload"], "host_ms_match_call": scope["host_ms_match_call"],
                "host_ms_tuples_out": scope["host_ms_tuples_out"],
                "bytes_up": scope["bytes_up"], "bytes_down": scope["bytes_down"],
                "tuple_form": ("compact: u32 word %s + one count byte per read" % scope["cbits"] if scope["form"] == "compact" else
                               "8-byte packed words %s" % scope["rbits"] if scope["form"] == "packed64" else "16-byte tuples"),
                "what": "the timed region of `value`: see config.timed_region",
            }
            match()
        # (3) a pass over freshly loaded reads with every buffer already allocated (what the CLI and
        # any service matching new batches pay per batch instead of the sized pass)
        if keep_reads and not args.unsorted:
            roff2 = synth.offsets_for(reads.shape[0], wl.read_len, device)
            eng.load_reads_device(reads.data_ptr(), roff2.data_ptr(), reads.shape[0])
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n = match()
            legs["first_pass_ms"] = (time.perf_counter() - t1) * 1e3
            legs["first_pass_device_ms"] = eng.stats()["ms_total"]
            assert n == n0
            del roff2
    if multi and scope is not None:
        legs["survey_scope"] = {
            "ms_per_pass": scope["ms_per_step"], "host_ms_queue_upload": scope["host_ms_queue_upload"],
            "host_ms_match_call": scope["host_ms_match_call"], "host_ms_tuples_out": scope["host_ms_tuples_out"],
            "bytes_up": scope["bytes_up"], "bytes_down": scope["bytes_down"], "host_slabs_pinned": scope["host_slabs_pinned"],
            "what": "rank 0's own figures (bytes per rank and step; host_ms_tuples_out = posting the gather + the copy of the rank's "
                    "buffer into its slab of rank 0's pinned host buffer): see config.timed_region",
        }
    legs["cold_pass_ms"] = cold_pass_ms
    legs["cold_pass_device_ms"] = cold_device_ms

    rc = 0
    # the workload's entry in TRAFFIC_FILE (runs with X have entries of their own: profiles/collect.sh cfg3xdb cfg3xreads)
    tkey = args.workload + (("xdb" if args.x_db_only else "xreads" if args.x_reads_only else "x") if args.xrate else "")
    tkey += "_classic" if (args.index == "classic" or (args.xrate and os.environ.get("MUSC_NO_X_CONTEXT"))
                           or os.environ.get("MUSC_CONTEXT") == "narrow") else ""
    if rank == 0:
        ms_step = elapsed * 1e3 / args.steps  # the steady-state pass (everything resident): `kernel_pipeline`, rooflines, per_step
        total_raw = wl.total_raw_reads if strong else wl.n_raw_reads * world
        value_resident = total_raw / (elapsed / args.steps)
        # `value`: SURVEY 8d's scope when it was timed (the default), else the resident pass (runs with X, --unsorted, --no-survey-scope)
        value = total_raw / (scope["elapsed"] / args.steps) if scope is not None else value_resident
        ms_value_step = scope["ms_per_step"] if scope is not None else ms_step
        L = wl.read_len
        rec_b = (2 * L + 7) // 8
        # k_confirm: what a launch loads -- one descriptor, one record, one target span per DESCRIPTOR
        # (a two-window descriptor is two of the reference's candidate pairs but is fetched once)
        ms_launch = acc["ms_confirm"] / max(acc["launches"], 1)
        bytes_launch = acc["bytes"] / max(acc["launches"], 1)
        achieved = (bytes_launch / 1e9) / (ms_launch / 1e3) if ms_launch > 0 else 0.0
        nl = max(st["confirm_launches"], 1)
        pair_bytes = (st["n_pairs"] * (12 + 2 * rec_b + 1) + 16 * st["n_hits"]) / nl
        confirm_roof = {
            "kernel": "k_confirm", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_peak": achieved / HBM_COPY_GBS,
            "traffic": measured_traffic(tkey, "k_confirm") if not args.reads else None,
            "traffic_source": TRAFFIC_FILE + " (rocprofv3 --pmc passes of this command, recorded; not measured by this run)",
            "algorithmic_bytes": "%d B per descriptor loaded (12 descriptor + %d read + %d target span) + 16 B per tuple written"
                                 % (12 + 2 * rec_b + 1, rec_b, rec_b + 1),
            "bytes_per_launch": bytes_launch,
            "descriptors_per_launch": st["n_descriptors"] / nl, "tuples_per_launch": st["n_hits"] / nl,
            "pairs_per_launch": st["n_pairs"] / nl,
            "achieved_pair_credited": (pair_bytes / 1e9) / (ms_launch / 1e3) if ms_launch > 0 else 0.0,
            "pair_credited_note": "r01's accounting: 63 B for each of the reference's candidate pairs, although a "
                                  "descriptor shared by two windows is loaded once; kept for comparison only",
            "avg_launch_ms": ms_launch, "launches_per_step": st["confirm_launches"],
        }
        # k_screen: records + bucket headers + index entries walked + descriptors written
        sl = max(st["n_batches"], 1)
        scr_bytes = (st["n_reads"] * rec_b + st["n_read_windows"] * 8 + st["n_candidates"] * 16 + st["n_descriptors"] * 12) / sl
        ms_scr = acc["ms_screen"] / max(acc["screen_launches"], 1)
        scr_ach = (scr_bytes / 1e9) / (ms_scr / 1e3) if ms_scr > 0 else 0.0
        lines = st["index_kind"] == 3  # line buckets: a probe is one 128-byte line (16 B header + 7 entries), walked by k_screen_t
        sname = "k_screen_t" if lines and os.environ.get("MUSC_SCREEN") != "wg" else "k_screen"
        screen_roof = {
            "kernel": sname, "bound": "hbm", "achieved": scr_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": scr_ach / HBM_PEAK_GBS, "frac_of_measured_copy_peak": scr_ach / HBM_COPY_GBS,
            "traffic": measured_traffic(tkey, sname) if not args.reads else None,
            "traffic_source": TRAFFIC_FILE + " (recorded PMC passes; not measured by this run)",
            "algorithmic_bytes": "%d B per read record + 8 B bucket header per probe + 16 B per index entry walked + "
                                 "12 B per descriptor written" % rec_b,
            "bytes_per_launch": scr_bytes, "probes_per_launch": st["n_read_windows"] / sl,
            "entries_per_launch": st["n_candidates"] / sl, "avg_launch_ms": ms_scr, "launches_per_step": st["n_batches"],
        }
        dominant = screen_roof if acc["ms_screen"] >= acc["ms_confirm"] else confirm_roof
        kind = st["index_kind"]
        match_roof = None
        if kind in (1, 2):
            # k_match (context buckets): screen + confirm + select in one kernel.  A launch loads every
            # read's record, ONE 128-byte bucket line per (read, window) probe -- the line carries the
            # placements' target bases, there is no target gather -- the overflow entries it walks, and
            # stages the tuples.
            mlc = max(acc["match_launches"], 1)
            ms_m = acc["ms_screen"] / mlc
            b_m, b_s = acc["match_bytes"] / mlc, acc["match_bytes_strict"] / mlc
            ach = (b_m / 1e9) / (ms_m / 1e3) if ms_m > 0 else 0.0
            ach_s = (b_s / 1e9) / (ms_m / 1e3) if ms_m > 0 else 0.0
            nlm = max(st["match_launches"], 1)
            # which of the fused kernels ran (musc_stats.match_variant): 2 / 3 k_match_t general / specialised
            # for the run's geometry, 4 / 5 k_match_g (three waves per SIMD, LDS-DMA) general / specialised
            mv = st.get("match_variant", 2)
            lane = True
            kname = {2: "k_match_t", 3: "k_match_t<geometry 1>", 4: "k_match_g", 5: "k_match_g<geometry 1>"}.get(mv, "k_match_t")
            fused_note = ("MUSC_FUSED_COMPACT=1: from the second launch of a pass on, a launch also moves the previous batch's staged "
                          "tuples into the hit list (32 B of traffic per tuple), which `achieved` does not bill"
                          if os.environ.get("MUSC_FUSED_COMPACT", "0") not in ("", "0") and st["match_launches"] > 1 else None)
            match_roof = {
                "kernel": kname + " (screen + confirm + per-read selection, context buckets)", "bound": "hbm",
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "frac_of_measured_copy_peak": ach / HBM_COPY_GBS,
                "traffic": measured_traffic(tkey, kname.split("<")[0]) if not args.reads else None,
                "traffic_source": TRAFFIC_FILE + " (rocprofv3 --pmc passes of this command, recorded; not measured by this run)",
                "algorithmic_bytes": "%d B per read record + 128 B bucket line per probe + %d B per overflow entry walked + "
                                     "16 B per tuple staged" % (rec_b, 60 if kind == 2 else 40),
                "bytes_per_launch": b_m,
                "achieved_strict": ach_s, "frac_strict": ach_s / HBM_PEAK_GBS,
                "strict_note": "a probe billed only for what it uses of its line: 8 B header + %d B per index entry present" % (60 if kind == 2 else 40),
                "reads_per_launch": st["n_reads"] / nlm, "probes_per_launch": st["n_read_windows"] / nlm,
                "entries_per_launch": st["n_candidates"] / nlm, "overflow_entries_per_launch": st["n_overflow_entries"] / nlm,
                "compared_per_launch": st["n_pairs"] / nlm, "tuples_per_launch": st["n_hits"] / nlm,
                "avg_launch_ms": ms_m, "launches_per_step": st["match_launches"], "not_billed": fused_note,
            }
            dominant, confirm_roof, screen_roof = match_roof, None, None
        for r_ in (dominant, confirm_roof, screen_roof):
            line_rate(r_)
        res = {
            "metric": "reads/sec (100 bp, multi-map) at 1/2/4/8 MI355X; confirm-kernel HBM GB/s vs peak",  # BASELINE.json's wording: value = reads/sec, the kernels are under "roofline*"
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_value_step, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {
                "workload": wl.name, "raw_reads_per_gpu": wl.n_raw_reads, "raw_reads_total": total_raw,
                "unique_reads_per_gpu": n_loaded,
                "targets": wl.n_targets, "target_len": wl.target_len, "read_len": wl.read_len,
                "Windows": list(wl.windows), "WindowWidth": wl.window_width, "PMatch": wl.pmatch,
                "MMTol": wl.mmtol, "MinDinuc": wl.min_dinuc, "MaxMatches": wl.max_matches,
                "MatchMode": wl.match_mode, "x_rate": args.xrate, "x_in": ("reads" if args.x_reads_only else "database" if args.x_db_only else "reads and database") if args.xrate else None, "read_order": "random" if args.unsorted else "bytewise sorted (reads_sorted)",
                "parallelism": "reads sharded x%d, database replicated" % world, "gather": gather_mode,
                "timed_region": (
                    "`value` and `ms_per_step`: SURVEY.md 8d's scope, EXACTLY `steps` steps between barrier + synchronize -- per step "
                    "each rank's packed unique reads (2 bits per base) in pinned host memory -> asynchronous upload in pieces -> "
                    "musc_match_device packs and matches each batch as its pieces arrive (a sizing pass: the reads are new to the "
                    "context) -> the tuples " +
                    ("into rank 0's host memory every step -- ONE pinned buffer owned by rank 0 with a slab per rank (shared memory, "
                     "registered with the HIP runtime by every rank), each rank copying its gather buffer into its slab over its own "
                     "PCIe link -- AND concatenated in rank 0's HBM over RCCL every step (HitGatherer: sizes agreed once, no host "
                     "synchronisation per step; the transfers of step i run while step i+1 uploads and matches), the last step's "
                     "concatenated slabs also copied from rank 0's HBM to its host inside the region" if multi else
                     "into pinned host memory (" + ("compact: u32 word + a count byte per read" if scope["form"] == "compact" else
                                                     "8-byte packed words" if scope["form"] == "packed64" else "16-byte tuples") + ")") +
                    "; database + index resident (uploaded once: one_off).  The pass with everything resident in HBM (r01-r03's `value`) "
                    "is `kernel_pipeline`: it is what `roofline` and `per_step` describe"
                ) if scope is not None else (
                    "`value`: steady state -- unique reads + database + index resident in HBM -> hits in HBM"
                    + (" gathered on rank 0 (RCCL, gather of pass i overlapped with pass i+1)" if world > 1 else "")
                    + ", repeat passes over the same reads (no sizing round trips): SURVEY 8d's scope was not timed in this run"),
            },
            "kernel_pipeline": {"ms_per_pass": ms_step, "reads_per_s": value_resident,
                                "what": "steady state: unique reads + database + index resident in HBM -> hits in HBM"
                                        + (" gathered on rank 0 (RCCL, gather of pass i overlapped with pass i+1)" if world > 1 else "")
                                        + ", repeat passes over the same reads (no sizing round trips), `steps` passes between barrier + synchronize"},
            "index": {"kind": "context buckets (128 B: 3 x 120 bases, fused k_match_t)" if kind == 1
                      else "wide context buckets (128 B: 2 x 200 bases, fused k_match_t)" if kind == 2
                      else "line buckets (128 B: header + 7 window starts; k_screen_t -> k_confirm)" if kind == 3
                      else "64-byte window-start buckets (k_screen -> k_confirm)",
                      "bytes": st["index_bytes"]},
            "roofline": dominant, "roofline_confirm": confirm_roof, "roofline_screen": screen_roof,
            "per_step": {
                "candidates": st["n_candidates"], "pairs": st["n_pairs"], "descriptors": st["n_descriptors"],
                "accepted": st["n_accepted"], "hits": st["n_hits"],
                "hits_on_rank0": gathered_n[0], "maxmatches_overflow_blocks": overflow_seen[0], "read_windows": st["n_read_windows"],
                "overflow_entries": st["n_overflow_entries"],
                "ms_screen": acc["ms_screen"] / args.steps, "ms_scan": acc["ms_scan"] / args.steps,
                "ms_confirm": acc["ms_confirm"] / args.steps,
                "ms_select": acc["ms_select"] / args.steps, "ms_device_total": acc["ms_total"] / args.steps,
                "host_overhead_ms": ms_step - acc["ms_total"] / args.steps, "ms_pass_wall": ms_step,
            },
            "one_off": {"db_pack_s": t_dbload, "index_build_ms": ms_index, "index_build_wall_s": t_index, **prep},
            **legs,
        }
        res["value_hbm_resident"] = value_resident  # (the number the bench contract's wording describes: inputs resident in HBM)
        if overflow_seen[0]:
            # a (window,key) block may hold more than MaxMatches accepted pairs on the union of the
            # shards: the tuples are a superset of the reference's until the truncation is replayed
            # (muscato_host.hpp apply_maxmatches does, on one GPU) -- not a valid measurement
            log("MaxMatches overflow suspected (%d): rerun on one GPU with the replay" % overflow_seen[0])
            res["value"] = None
            rc = 4
        if keep_for_cpu:
            try:
                res["cpu_baseline"] = cpu_baseline(wl, cfg, targets, reads, lambda: Engine(local_rank),
                                                   wl.n_raw_reads, log)
                if not res["cpu_baseline"]["gpu_bit_exact_on_sample"]:
                    res["value"] = None  # a fast pass with wrong tuples is not a measurement
                    rc = 5
            except Exception as e:  # the baseline must never hide the measurement
                res["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(res), flush=True)
    eng.close()
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return rc


if __name__ == "__main__":
    sys.exit(main())
