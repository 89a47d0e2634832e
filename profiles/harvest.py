#!/usr/bin/env python3
"""Copy a round's summaries from gpurun_out/ (scratch) into profiles/ (tracked):
    python3 profiles/harvest.py r04 [a b c d]
bench lines (the JSON line only), rehearsals, the kernel statistics and under-rocprof lines of the collection
calls (tags r04a ... renamed to r04), their traffic records merged into one <tag>_traffic.json keyed by workload,
counter summaries, fuzz totals, whole-workload CPU comparisons."""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def json_line(src, dst):
    try:
        lines = [l for l in open(src) if l.startswith("{")]
    except OSError:
        return False
    if not lines:
        return False
    open(dst, "w").write(lines[-1])
    return True


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    parts = sys.argv[2:] or ["a", "b", "c", "d"]
    n = 0
    for f in sorted(glob.glob(os.path.join(G, tag + "_*_bench.json"))) + \
            [os.path.join(G, tag + "_n2_rehearsal.json"), os.path.join(G, tag + "_rccl1_rehearsal.json")] + \
            sorted(glob.glob(os.path.join(G, tag + "_cpu_full*.json"))):
        n += json_line(f, os.path.join(P, os.path.basename(f)))
    traffic = {}
    for part in parts:
        for f in sorted(glob.glob(os.path.join(G, tag + part + "_*_kernel_stats.csv"))):
            shutil.copy(f, os.path.join(P, os.path.basename(f).replace(tag + part + "_", tag + "_")))
            n += 1
        for f in sorted(glob.glob(os.path.join(G, tag + part + "_*_bench_under_rocprof.json"))):
            n += json_line(f, os.path.join(P, os.path.basename(f).replace(tag + part + "_", tag + "_")))
        try:
            t = json.load(open(os.path.join(G, tag + part + "_traffic.json")))
        except OSError:
            continue
        method = t.pop("method", None)
        traffic.update(t)
        if method:
            traffic["method"] = method
    if traffic:
        json.dump(traffic, open(os.path.join(P, tag + "_traffic.json"), "w"), indent=1)
        n += 1
    for src, dst in (("pmc_sq_t.txt", "_cfg3_sq_t.txt"), ("pmc_mem_t.txt", "_cfg3_pmc_mem_t.txt"), ("pmc_sq_s5.txt", "_cfg5_sq_k_screen_t.txt"),
                     (tag + "_fuzz_totals.txt", "_fuzz_totals.txt")):
        if os.path.exists(os.path.join(G, src)):
            shutil.copy(os.path.join(G, src), os.path.join(P, tag + dst))
            n += 1
    print("%d files into profiles/" % n)


if __name__ == "__main__":
    main()
