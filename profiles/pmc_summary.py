#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (sum / per-dispatch mean)."""
import csv
import glob
import sys
from collections import defaultdict


def main(root):
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if not k.startswith("k_"):
                continue
            a = agg[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    for k in sorted(agg):
        print(k)
        for c in sorted(agg[k]):
            s, n = agg[k][c]
            print("    %-36s dispatches=%4d sum=%.4g mean=%.4g" % (c, n, s, s / n))


if __name__ == "__main__":
    main(sys.argv[1])
