#!/bin/bash
# cfg3 device times of library variants: profiles/variants.sh [bench flags --] build_variants/a.so build_variants/b.so ...
flags=""
if [ "$1" = "--flags" ]; then flags="$2"; shift 2; fi
for lib in "$@"; do
  MUSC_LIB_PATH=$PWD/$lib timeout -k 10 200 python bench.py --workload cfg3 --no-cpu-baseline --no-survey-scope --steps 10 $flags > gpurun_out/var.json 2> gpurun_out/var.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/var.json') if l.startswith('{')][-1])
    p=d['per_step']
    print('$lib', 'ms/pass %.3f' % d['ms_per_step'], 'k_match/launch %.3f' % (p['ms_screen']/3), 'select %.3f' % p['ms_select'], 'hits', p['hits'])
except Exception as e: print('$lib failed', e)
PY
done
