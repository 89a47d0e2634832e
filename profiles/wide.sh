#!/bin/bash
# the wide-bucket workloads on the GPU box: device times on wide context buckets and, beside them, the
# same runs on the two-kernel path (what they took before wide buckets existed)
#   profiles/wide.sh
set -o pipefail
for wl in cfg3w3 cfg3r150; do
for mode in auto classic; do
  if [ $mode = classic ]; then export MUSC_CONTEXT=narrow; else unset MUSC_CONTEXT; fi
  timeout -k 10 400 python bench.py --workload $wl --no-cpu-baseline --no-survey-scope --steps 10 > gpurun_out/wide_${wl}_$mode.json 2> gpurun_out/wide_${wl}_$mode.err || { tail -5 gpurun_out/wide_${wl}_$mode.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/wide_${wl}_$mode.json') if l.startswith('{')][-1])
p=d['per_step']
print('$mode $wl', d['index']['kind'], 'ms/pass %.3f' % d['ms_per_step'], 'screen|match %.3f' % p['ms_screen'], 'confirm %.3f' % p['ms_confirm'], 'device %.3f' % p['ms_device_total'], 'hits', p['hits'], 'Mreads/s %.1f' % (d['value'] / 1e6), 'roofline', d['roofline'].get('frac'))
PY
done
done
