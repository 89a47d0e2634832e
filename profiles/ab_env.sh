#!/bin/bash
# A/B of environment knobs on one box: profiles/ab_env.sh [--wl cfg3] "VAR=val" "" "VAR=val" ""   (an empty string = the default)
wl=cfg3
if [ "$1" = "--wl" ]; then wl="$2"; shift 2; fi
for kv in "$@"; do
  timeout -k 10 300 env $kv python bench.py --workload $wl --no-cpu-baseline --no-survey-scope --steps 10 > gpurun_out/abenv.json 2> gpurun_out/abenv.err
  python - <<PY
import json
try:
    d=json.loads([l for l in open('gpurun_out/abenv.json') if l.startswith('{')][-1])
    p=d['per_step']; r=d['roofline']; rc=d.get('roofline_confirm') or {}; rs=d.get('roofline_screen') or {}
    print('[%s]' % '$kv', '$wl', 'ms/pass %.3f' % d['ms_per_step'], '|', r['kernel'].split(' ')[0], '%.4f ms/launch' % r['avg_launch_ms'], 'frac %.3f' % r['frac'],
          '|', ' '.join('%s %.4f ms frac %.3f' % (x['kernel'], x['avg_launch_ms'], x['frac']) for x in (rs, rc) if x and x is not r and x.get('kernel') != r.get('kernel')), '| hits', p['hits'])
except Exception as e: print('[%s] failed' % '$kv', e)
PY
done
