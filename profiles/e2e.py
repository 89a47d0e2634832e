#!/usr/bin/env python3
"""End-to-end wall time of the file-level drop-in (FASTQ + target file -> results.txt), reported
beside the device-pipeline metric as SURVEY.md 8(d) asks.  Synthetic data of BASELINE configs[1]
shape (reads sampled from the targets with 1 % substitutions, 10 % duplicates, 20 % random).
usage: e2e.py <workdir> [n_reads] [n_targets]"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "muscato_amd", "bin")


def main():
    wd = sys.argv[1]
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
    n_targets = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
    L, TL = 100, 1000
    os.makedirs(wd, exist_ok=True)
    rng = np.random.default_rng(1)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    T = lut[rng.integers(0, 4, size=(n_targets, TL), dtype=np.uint8)]
    t0 = time.time()
    with open(os.path.join(wd, "genes.txt"), "wb") as f:
        for i in range(n_targets):
            f.write(b"gene%d\t" % i + T[i].tobytes() + b"\n")
    g = rng.integers(0, n_targets, size=n_reads)
    p = rng.integers(0, TL - L + 1, size=n_reads)
    R = T[g[:, None], p[:, None] + np.arange(L)[None, :]].copy()
    sub = rng.random(R.shape) < 0.01
    R[sub] = lut[rng.integers(0, 4, size=int(sub.sum()))]
    rnd = rng.random(n_reads) < 0.2
    R[rnd] = lut[rng.integers(0, 4, size=(int(rnd.sum()), L))]
    dup = rng.random(n_reads) < 0.1
    R[dup] = R[rng.integers(0, n_reads, size=int(dup.sum()))]
    qual = b"F" * L
    with open(os.path.join(wd, "reads.fastq"), "wb") as f:
        for i in range(n_reads):
            f.write(b"@read%d\n" % i + R[i].tobytes() + b"\n+\n" + qual + b"\n")
    t_gen = time.time() - t0
    t0 = time.time()
    subprocess.check_call([os.path.join(BIN, "muscato_prep_targets"), "genes.txt"], cwd=wd)
    t_prep = time.time() - t0
    cfg = {"ReadFileName": "reads.fastq", "GeneFileName": "musc_genes.txt.sz", "GeneIdFileName": "musc_ids_genes.txt.sz",
           "ResultsFileName": "results.txt", "Windows": [0, 20], "WindowWidth": 15, "PMatch": 0.97, "MinDinuc": 5,
           "MaxReadLength": 100, "MaxMatches": 1000000, "MMTol": 0, "MatchMode": "best"}
    with open(os.path.join(wd, "config.json"), "w") as f:
        json.dump(cfg, f)
    out = {}
    t0 = time.time()
    r = subprocess.run([os.path.join(BIN, "muscato"), "-ConfigFileName=config.json"], cwd=wd, stderr=subprocess.PIPE)
    out["muscato_wall_s"] = round(time.time() - t0, 2)
    if r.returncode:
        sys.stderr.write(r.stderr.decode())
        return 1
    logs = sorted((os.path.join(wd, "muscato_logs", d) for d in os.listdir(os.path.join(wd, "muscato_logs"))),
                  key=os.path.getmtime)
    for lg in logs[-1:]:
        sys.stderr.write(open(os.path.join(lg, "muscato.log")).read())
    with open(os.path.join(wd, "results.txt"), "rb") as f:
        nres = sum(1 for _ in f)
    out.update({"reads": n_reads, "targets": n_targets, "result_lines": nres, "generate_s": round(t_gen, 1),
                "prep_targets_s": round(t_prep, 2),
                "fastq_bytes": os.path.getsize(os.path.join(wd, "reads.fastq")),
                "results_bytes": os.path.getsize(os.path.join(wd, "results.txt"))})
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
