#!/usr/bin/env python3
"""From a rocprofv3 --memory-copy-trace --kernel-trace run of bench.py: how much of the host-to-device
copies of the SURVEY-scope leg (musc_reads_load_packed32 async: the read upload in pieces) ran while a
match kernel was executing.  usage: overlap_from_trace.py <rocprofv3 output dir>"""
import csv
import glob
import sys


def intervals(path, name_col, want):
    out = []
    for r in csv.DictReader(open(path)):
        if want(r.get(name_col, "")):
            out.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get(name_col, "")))
    return sorted(out)


def main(root):
    kt = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
    mt = glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True)
    if not kt or not mt:
        print("no traces under", root)
        return
    ks = intervals(kt[0], "Kernel_Name", lambda n: "k_match" in n or "k_pack_reads_fixed" in n)
    rows = list(csv.DictReader(open(mt[0])))
    cps = []
    for r in rows:
        d = r.get("Direction", r.get("Kind", ""))
        cps.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), d, int(r.get("Size", 0) or 0)))
    cps.sort()
    big = [c for c in cps if c[1] - c[0] > 200_000 and "HOST_TO_DEVICE" in c[2].upper().replace("MEMORY_COPY_", "")]
    tot = sum(e - s for s, e, _, _ in big)
    ov = 0
    for s, e, _, _ in big:
        for ks_, ke, _ in ks:
            lo, hi = max(s, ks_), min(e, ke)
            if hi > lo:
                ov += hi - lo
    print("host-to-device copies longer than 0.2 ms: %d, %.2f ms in total (%.2f GB); of that %.2f ms (%.0f %%) while a "
          "k_match* / k_pack_reads_fixed kernel was executing" % (len(big), tot / 1e6, sum(c[3] for c in big) / 1e9, ov / 1e6,
                                                                    100.0 * ov / max(tot, 1)))
    # the last SURVEY-scope repetition as a time line: copies and match kernels, ms from its first copy
    if big:
        t0 = big[-12][0] if len(big) >= 12 else big[0][0]
        print("time line of the last repetition (ms from its first upload piece):")
        ev = [(s, e, "H2D %.0f MB" % (sz / 1e6)) for s, e, _, sz in big if s >= t0] + \
             [(s, e, n.split("(")[0].replace("void ", "")[:40]) for s, e, n in ks if s >= t0]
        for s, e, n in sorted(ev)[:60]:
            print("  %8.3f - %8.3f  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, n))


if __name__ == "__main__":
    main(sys.argv[1])
