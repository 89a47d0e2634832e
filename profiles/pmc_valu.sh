#!/bin/bash
# dynamic instruction counts of the match kernel per library variant:
#   profiles/pmc_valu.sh product build_variants/a.so ...   (one --pmc pass per variant)
export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = "product" ]; then unset MUSC_LIB_PATH; else export MUSC_LIB_PATH=$PWD/$lib; fi
  out=gpurun_out/pmc_v; rm -rf $out; mkdir -p $out
  timeout -k 10 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d $out/p -- python3 bench.py --workload cfg3 --no-cpu-baseline --no-survey-scope --steps 1 --warmup 1 > $out/p.log 2>&1 || { echo "$lib failed"; tail -3 $out/p.log; continue; }
  python3 - "$lib" <<PY
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob('gpurun_out/pmc_v/p/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_match' not in k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); 
        if r['Counter_Name'] == 'SQ_WAVES': n[k] += 1
for k in acc:
    tiles = 233550.0  # wave-tiles per cfg3 launch (14.95 M reads / 64)
    d = acc[k]; m = max(n[k], 1)
    print(sys.argv[1], k[:28], 'launches', m, 'per wave-tile: VALU %.0f SALU %.0f LDS %.0f' % (d['SQ_INSTS_VALU']/m/tiles, d['SQ_INSTS_SALU']/m/tiles, d['SQ_INSTS_LDS']/m/tiles))
PY
done
