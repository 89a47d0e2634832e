#!/bin/bash
# A/B of environment knobs on SURVEY 8d's scope (`value`) and the resident pass, one box: profiles/ab_scope.sh [--wl cfg3] "VAR=val" "" ...
wl=cfg3
if [ "$1" = "--wl" ]; then wl="$2"; shift 2; fi
for kv in "$@"; do
  timeout -k 10 300 env $kv python bench.py --workload $wl --no-cpu-baseline --steps 10 > gpurun_out/abscope.json 2> gpurun_out/abscope.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/abscope.json') if l.startswith('{')][-1])
    s=d['survey_scope']; kp=d['kernel_pipeline']; r=d['roofline']
    print('[%s]' % '$kv', '$wl', 'value: %.3f ms/step' % d['ms_per_step'], '| serial up %.2f match %.2f down %.2f' % (s['serial']['ms_upload_then_pack'], s['serial']['ms_match'], s['serial']['ms_tuples_to_host']),
          '| resident pass %.3f ms' % kp['ms_per_pass'], '|', r['kernel'].split(' ')[0], '%.4f ms/launch x %d' % (r['avg_launch_ms'], r['launches_per_step']), '| hits', d['per_step']['hits'])
except Exception as e: print('[%s] failed' % '$kv', e)
PY
done
