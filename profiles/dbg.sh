#!/bin/bash
# k_match with parts switched off (MUSC_DEBUG_MATCH: 1 no comparisons, 2 no bucket loads, 4 no overflow lists)
for d in ${DBGS:-0 1 2 3 4 5 7}; do
  MUSC_DEBUG_MATCH=$d timeout -k 10 200 python bench.py --workload cfg3 --no-cpu-baseline --no-survey-scope --steps 5 > gpurun_out/dbg_$d.json 2> gpurun_out/dbg_$d.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/dbg_$d.json') if l.startswith('{')][-1])
    p=d['per_step']
    print('dbg $d', 'k_match/launch %.3f' % (p['ms_screen']/3), 'select %.3f' % p['ms_select'], 'hits', p['hits'])
except Exception as e: print('dbg $d failed', e)
PY
done
