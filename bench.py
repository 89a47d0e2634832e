#!/usr/bin/env python3
"""bench.py -- reads/sec of muscato's seed-and-extend hot path (screen + confirm + per-read
best filter) on MI355X, with the confirm kernel's HBM roofline and a CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3|cfg2|...]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic unique reads that are already
resident in HBM, against the resident target database + k-mer index; hits stay on the device
(N=1) or are concatenated on rank 0 over RCCL (N>1, inside the timed region).  Scaling is weak:
every rank processes its own shard of `n_raw_reads` raw reads against the replicated database.
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


def measured_traffic(workload_key):
    """HBM bytes per k_confirm launch from the PMC passes recorded under profiles/ (rocprofv3
    --pmc cannot run inside this process); None if no record matches the workload."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            return json.load(f)[workload_key]["traffic_bytes_per_launch"]
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
    single = {"value": n_raw_full / t_full1 if t_full1 > 0 else 0.0, "cores": 1,
              "sample": "%d reads x %d targets, same port and extrapolation, %.1fs wall" % (s1, tp1, wall1)}

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
    return {
        "value": value, "unit": "reads/s", "cores": nthr, "kind": "port",
        "sample": ("%d evenly spaced unique reads (=%d raw) x first %d of %d targets through oracle/literal.cpp "
                   "(NumHash=20, BloomSize=4e9): scan %.2fs, windows+bloom %.2fs, candidate sort+confirm %.2fs; "
                   "extrapolated to the full batch as scan*%.0f + read terms*%.1f + pair terms*%.1f*%.0f = %.1fs"
                   % (s, s + s // 9, tp, T, t_scan, t_win + t_bloom, t_csort + t_conf, ft, fr, fr, ft, t_full)),
        "sample_wall_s": wall, "gpu_bit_exact_on_sample": exact, "sample_hits": int(len(exp)),
        "single_thread": single,
    }


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3")
    ap.add_argument("--reads", type=int, default=0, help="override raw reads per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unsorted", action="store_true", help="leave the unique reads in random order")
    ap.add_argument("--no-block-check", action="store_true", help="skip the MaxMatches per-block overflow check")
    ap.add_argument("--xrate", type=float, default=0.0,
                    help="fraction of bases replaced by X in targets and reads (the correctness/timing run with the mask planes)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print("bench.py: --gpus %d needs a torch.distributed.run launch with that many ranks" % args.gpus,
                  file=sys.stderr)
            return 2
        args.gpus = world

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
    if "MUSC_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["MUSC_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    wl = synth.WORKLOADS[args.workload]
    if args.reads:
        wl = synth.Workload(**{**wl.__dict__, "n_raw_reads": args.reads,
                               "name": wl.name + " (reads=%d)" % args.reads})
    cfg = Config(Windows=list(wl.windows), WindowWidth=wl.window_width, PMatch=wl.pmatch,
                 MinDinuc=wl.min_dinuc, MaxReadLength=wl.read_len, MaxMatches=wl.max_matches,
                 MMTol=wl.mmtol, MatchMode=wl.match_mode)
    seed = synth.SEED_BASE + sum(ord(c) for c in args.workload)

    t0 = time.time()
    targets = synth.gen_targets(wl, device, seed)
    if args.xrate > 0:
        gx = torch.Generator(device=device)
        gx.manual_seed(seed + 99)
        for s0 in range(0, wl.n_targets, 100_000):
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
    eng.build_index(wl.window_width)
    t_index = time.time() - t0
    ms_index = eng.stats()["ms_index_build"]
    log("db pack %.3fs, index build %.3fs (device %.1f ms)" % (t_dbload, t_index, ms_index))

    t0 = time.time()
    U = wl.n_unique_reads
    reads = synth.gen_unique_reads(wl, targets, device, seed + 7919 * (rank + 1))
    if args.xrate > 0:
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
        n_distinct = int((reads[1:] != reads[:-1]).any(dim=1).sum()) + 1
        prep["equals_torch_sort"] = bool(increasing and n_distinct == len(ustart) - 1)
        del raw, raw_off, heads, order, ustart
        log("read prep (sort + collapse) of %d raw reads -> %d distinct: %.1f ms on the device, %.2f s wall incl. "
            "the 200 MB order download; sorted, distinct and as many as torch's sort finds: %s"
            % (n_raw, prep["distinct"], prep["read_prep_device_ms"], prep["read_prep_wall_s"], prep["equals_torch_sort"]))

    keep_for_cpu = (rank == 0 and world == 1 and not args.no_cpu_baseline)
    if not keep_for_cpu:
        del reads, targets
    del roff, toff
    torch.cuda.empty_cache()

    read_base = rank * U
    gathered_n = [0]
    gatherer = None
    if world > 1:
        # The tuples of every pass end up on rank 0 in rank order (= global read order).  The
        # gather of pass i (one collective, counts ride in the buffers) runs on the
        # communicator's stream while pass i+1 is matched; capacity agreed once, untimed.
        from muscato_amd.dist import HitGatherer
        n0 = eng.match_device(cfg, apply_mmtol=True, skip_block_check=args.no_block_check)
        gdev = device if backend == "nccl" else torch.device("cpu")
        # 8-byte tuples on the links when the fields fit 64 bits: global read number, target
        # number, position, mismatch count (rank 0 ingests 7 shards' tuples per pass at 8 GPUs)
        budget = int((1.0 - wl.pmatch) * wl.read_len)
        pack_bits = [max(1, (world * U - 1).bit_length()), max(1, (wl.n_targets - 1).bit_length()),
                     max(1, wl.target_len.bit_length()), max(1, budget.bit_length())]
        use_packed = sum(pack_bits) <= 64 and not os.environ.get("MUSC_BENCH_UNPACKED")
        gatherer = HitGatherer(HitGatherer.agree_capacity(n0, gdev), gdev, packed=use_packed)

    gather_mode = ["none" if world == 1 else
                   "overlapped (HitGatherer), " + ("8-byte packed tuples %s" % pack_bits if use_packed else "16-byte tuples")]

    def step():
        n = eng.match_device(cfg, apply_mmtol=True, skip_block_check=args.no_block_check)
        if world > 1 and gather_mode[0].startswith("overlapped"):
            def fill(buf):
                if n and gatherer.packed:
                    eng.hits_to_packed(buf.data_ptr(), n, buf.is_cuda, pack_bits, read_base)
                elif n:
                    eng.hits_to(buf.data_ptr(), n, buf.is_cuda)
                return n
            try:
                gatherer.submit(fill, read_base)
                return n
            except Exception as e:  # same tuples on rank 0, just without the overlap
                log("overlapped gather failed (%r); falling back to one synchronous gather per pass" % (e,))
                gather_mode[0] = "synchronous (gather_hits)"
        if world > 1:
            from muscato_amd.dist import gather_hits
            h = torch.empty((max(n, 1), 4), dtype=torch.int32, device=device)
            if n:
                eng.hits_to(h.data_ptr(), n, True)
            g_all = gather_hits(h[:n] if backend == "nccl" else h[:n].cpu(), read_base, dst=0)
            if g_all is not None:
                gathered_n[0] = int(g_all.shape[0])
        else:
            gathered_n[0] = n
        return n

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    acc = {"ms_confirm": 0.0, "launches": 0, "bytes": 0, "ms_screen": 0.0, "ms_scan": 0.0,
           "ms_select": 0.0, "ms_total": 0.0}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        st = eng.stats()  # host-side read of numbers the library already holds
        acc["ms_confirm"] += st["ms_confirm"]; acc["launches"] += st["confirm_launches"]
        acc["bytes"] += st["confirm_bytes"]
        for k in ("ms_screen", "ms_scan", "ms_select", "ms_total"):
            acc[k] += st[k]
    if gatherer is not None and gather_mode[0].startswith("overlapped"):
        cnts = gatherer.finish()  # the last gathers complete inside the timed region
        if cnts is not None:
            gathered_n[0] = sum(cnts)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = eng.stats()

    # PCIe-inclusive variant (never `value`): one extra step that also copies the hits to the host
    pcie_ms = None
    if world == 1:
        import numpy as np
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n = eng.match_device(cfg, apply_mmtol=True)
        out = np.empty((max(n, 1), 4), dtype=np.uint32)
        if n:
            eng.hits_to(out.ctypes.data, n, False)
        pcie_ms = (time.perf_counter() - t1) * 1e3
        del out

    if rank == 0:
        ms_step = elapsed * 1e3 / args.steps
        total_raw = wl.n_raw_reads * world
        value = total_raw / (elapsed / args.steps)
        ms_launch = acc["ms_confirm"] / max(acc["launches"], 1)
        bytes_launch = acc["bytes"] / max(acc["launches"], 1)
        achieved = (bytes_launch / 1e9) / (ms_launch / 1e3) if ms_launch > 0 else 0.0
        res = {
            "metric": "reads/sec (100 bp, multi-map) at 1/2/4/8 MI355X; confirm-kernel HBM GB/s vs peak",  # BASELINE.json's wording: value = reads/sec, the confirm kernel is under "roofline"
            
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {
                "workload": wl.name, "raw_reads_per_gpu": wl.n_raw_reads, "unique_reads_per_gpu": eng.n_reads,
                "targets": wl.n_targets, "target_len": wl.target_len, "read_len": wl.read_len,
                "Windows": list(wl.windows), "WindowWidth": wl.window_width, "PMatch": wl.pmatch,
                "MMTol": wl.mmtol, "MinDinuc": wl.min_dinuc, "MaxMatches": wl.max_matches,
                "MatchMode": wl.match_mode, "x_rate": args.xrate, "read_order": "random" if args.unsorted else "bytewise sorted (reads_sorted)",
                "parallelism": "reads sharded x%d, database replicated" % world, "gather": gather_mode[0],
                "timed_region": "unique reads + database + index resident in HBM -> hits in HBM"
                                + (" gathered on rank 0 (RCCL, gather of pass i overlapped with pass i+1)" if world > 1 else ""),
            },
            "roofline": {
                "kernel": "k_confirm", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_peak": achieved / HBM_COPY_GBS,
                "traffic": measured_traffic(args.workload) if not args.reads else None,
                "algorithmic_bytes": "63 B per candidate pair (12 descriptor + 25 read + 26 target span at 100 bp) + 16 B per tuple written",
                "pairs_per_launch": st["n_pairs"] / max(st["confirm_launches"], 1),
                "descriptors_per_launch": st["n_descriptors"] / max(st["confirm_launches"], 1),
                "tuples_per_launch": st["n_hits"] / max(st["confirm_launches"], 1),
                "avg_launch_ms": ms_launch, "launches_per_step": st["confirm_launches"],
            },
            "per_step": {
                "candidates": st["n_candidates"], "pairs": st["n_pairs"], "descriptors": st["n_descriptors"],
                "accepted": st["n_accepted"], "hits": st["n_hits"],
                "hits_on_rank0": gathered_n[0], "maxmatches_overflow_blocks": st["n_overflow_blocks"], "read_windows": st["n_read_windows"],
                "ms_screen": acc["ms_screen"] / args.steps, "ms_scan": acc["ms_scan"] / args.steps,
                "ms_confirm": acc["ms_confirm"] / args.steps,
                "ms_select": acc["ms_select"] / args.steps, "ms_device_total": acc["ms_total"] / args.steps,
            },
            "one_off": {"db_pack_s": t_dbload, "index_build_ms": ms_index, **prep},
        }
        if pcie_ms is not None:
            res["pcie_inclusive"] = {"ms_per_step": pcie_ms, "reads_per_s": wl.n_raw_reads / (pcie_ms / 1e3)}
        if keep_for_cpu:
            try:
                res["cpu_baseline"] = cpu_baseline(wl, cfg, targets, reads, lambda: Engine(local_rank),
                                                   wl.n_raw_reads, log)
            except Exception as e:  # the baseline must never hide the measurement
                res["cpu_baseline"] = {"value": None, "unit": "reads/s", "cores": 0, "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        print(json.dumps(res), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
