#!/bin/bash
# the match kernel with parts switched off: the experiment knobs are compile-time (-DMUSC_LANE_DBG=n for k_match_t:
# 1 no comparisons, 2 no bucket loads, 4 no overflow lists, 8 no phase D ...), so every value is a library variant
for d in ${DBGS:-0 1 2 4}; do
  if [ "$d" = 0 ]; then unset MUSC_LIB_PATH; else
    python -m muscato_amd.build variant dbg$d -DMUSC_LANE_DBG=$d > /dev/null && export MUSC_LIB_PATH=$PWD/build_variants/dbg$d.so
  fi
  timeout -k 10 200 python bench.py --workload cfg3 --no-cpu-baseline --no-survey-scope --steps 5 > gpurun_out/dbg_$d.json 2> gpurun_out/dbg_$d.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/dbg_$d.json') if l.startswith('{')][-1])
    p=d['per_step']
    print('dbg $d', 'k_match/launch %.3f' % (p['ms_screen']/3), 'select %.3f' % p['ms_select'], 'hits', p['hits'])
except Exception as e: print('dbg $d failed', e)
PY
done
