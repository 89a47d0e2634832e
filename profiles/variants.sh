#!/bin/bash
# cfg3 device times of library variants: profiles/variants.sh [--flags "bench flags"] [--wl cfg3] build_variants/a.so ...
flags=""; wl=cfg3
while [ "$1" = "--flags" ] || [ "$1" = "--wl" ]; do
  if [ "$1" = "--flags" ]; then flags="$2"; else wl="$2"; fi
  shift 2
done
for lib in "$@"; do
  if [ "$lib" = "product" ]; then unset MUSC_LIB_PATH; else export MUSC_LIB_PATH=$PWD/$lib; fi
  timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-survey-scope --steps 10 $flags > gpurun_out/var.json 2> gpurun_out/var.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/var.json') if l.startswith('{')][-1])
    p=d['per_step']
    print('$lib', 'ms/pass %.3f' % d['ms_per_step'], 'k_match/launch %.3f' % (p['ms_screen']/max(1,p.get('match_launches',3))), 'select %.3f' % p['ms_select'], 'hits', p['hits'])
except Exception as e: print('$lib failed', e)
PY
done
