#!/usr/bin/env python3
"""The roofline table of DESIGN.md 5 / BASELINE.md from a round's committed files:
    python3 profiles/tables.py r04
reads profiles/<tag>_<workload>_bench.json (HIP-event launch times, algorithmic bytes),
profiles/<tag>_<workload>_auto_kernel_stats.csv + ..._bench_under_rocprof.json (rocprofv3 averages and that
run's own HIP-event figure) and profiles/<tag>_traffic.json (PMC bytes per launch); prints markdown rows."""
import csv
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def last_json(path):
    try:
        return json.loads([l for l in open(path) if l.startswith("{")][-1])
    except Exception:
        return None


def kstats(path, prefix):
    """(calls, average ms, min ms, max ms) of the kernel rows whose name starts with `prefix`, calls-weighted"""
    try:
        rows = [r for r in csv.DictReader(open(path)) if r["Name"].replace("void ", "").startswith(prefix)]
    except Exception:
        return None
    if not rows:
        return None
    calls = sum(int(r["Calls"]) for r in rows)
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    return calls, tot / calls / 1e6, min(float(r["MinNs"]) for r in rows) / 1e6, max(float(r["MaxNs"]) for r in rows) / 1e6


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    traffic = last_json(os.path.join(HERE, tag + "_traffic.json")) or {}
    try:
        traffic = json.load(open(os.path.join(HERE, tag + "_traffic.json")))
    except Exception:
        traffic = {}
    wls = [("cfg3", "cfg3", "auto"), ("cfg4shard", "cfg4shard", "auto"), ("cfg2", "cfg2", "auto"), ("cfg3_dma", "cfg3", "dma"),
           ("cfg3w3", "cfg3w3", "auto"), ("cfg3r150", "cfg3r150", "auto"), ("cfg3_xreads", "cfg3xreads", "auto"),
           ("cfg3_xdb", "cfg3xdb", "auto"), ("cfg3_classic", "cfg3", "classic"), ("cfg5shard", "cfg5shard", "auto"),
           ("bigtest", "bigtest", "auto")]
    print("| workload | kernel | ms per launch (HIP events, bench line) | rocprofv3: calls, average [min .. max] (HIP events of that run) | bytes billed per launch | of 8 TB/s (HIP events / rocprofv3 average) | PMC traffic per launch |")
    print("|---|---|---|---|---|---|---|")
    for bench_key, prof_key, kind in wls:
        d = last_json(os.path.join(HERE, "%s_%s_bench.json" % (tag, bench_key)))
        if d is None:
            continue
        roofs = [d["roofline"]]
        if d.get("roofline_confirm") and d["roofline_confirm"].get("kernel") != d["roofline"].get("kernel"):
            roofs.append(d["roofline_confirm"])
        if d.get("roofline_screen") and d["roofline_screen"].get("kernel") != d["roofline"].get("kernel"):
            roofs.insert(0, d["roofline_screen"])
        for r in roofs:
            kname = r["kernel"].split(" ")[0].split("<")[0]
            st = kstats(os.path.join(HERE, "%s_%s_%s_kernel_stats.csv" % (tag, prof_key, kind)), kname)
            under = last_json(os.path.join(HERE, "%s_%s_%s_bench_under_rocprof.json" % (tag, prof_key, kind if kind != "dma" else "auto")))
            under_ms = None
            if under and kind != "dma":
                for rr in (under.get("roofline"), under.get("roofline_confirm"), under.get("roofline_screen")):
                    if rr and rr["kernel"].split(" ")[0].split("<")[0] == kname:
                        under_ms = rr["avg_launch_ms"]
            tkey = prof_key + ("_classic" if kind == "classic" else "")
            tr = (traffic.get(tkey) or {}).get(kname)
            by = r["bytes_per_launch"]
            frac_prof = by / (st[1] * 1e-3) / 8e12 if st else None
            print("| %s | `%s` | %.3f | %s | %.3f GB | %.3f / %s | %s |" % (
                bench_key, r["kernel"].split(" (")[0], r["avg_launch_ms"],
                ("%d, %.4f [%.3f .. %.3f]%s" % (st[0], st[1], st[2], st[3], (" (%.3f)" % under_ms) if under_ms else "")) if st else "-",
                by / 1e9, r["frac"], ("%.3f" % frac_prof) if frac_prof else "-",
                ("%.3f GB = %.2f x; L2 hit rate %.2f" % (tr["traffic_bytes_per_launch"] / 1e9, tr["traffic_bytes_per_launch"] / by, tr.get("l2_hit_rate", float("nan")))) if tr else "-"))
    print()
    print("| workload | ms per step on SURVEY 8d's scope (`value`) | reads/s | resident pass (`kernel_pipeline`) ms | reads/s |")
    print("|---|---|---|---|---|")
    for bench_key, _, _ in wls:
        d = last_json(os.path.join(HERE, "%s_%s_bench.json" % (tag, bench_key)))
        if d is None:
            continue
        kp = d.get("kernel_pipeline") or {}
        sc = d.get("survey_scope")
        print("| %s | %s | %s | %.3f | %.3g |" % (bench_key, ("%.2f" % d["ms_per_step"]) if sc else "-", ("%.3g" % d["value"]) if sc else "-",
                                                  kp.get("ms_per_pass", d["ms_per_step"]), kp.get("reads_per_s", d["value"])))


if __name__ == "__main__":
    main()
