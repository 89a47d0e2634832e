#!/bin/bash
# quick check on the GPU box: parity tests of the kernels, then cfg3 and cfg2 device times
#   profiles/quick.sh [notest]
set -o pipefail
if [ "$1" != "notest" ]; then
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/quick_tests.log 2>&1 || { tail -30 gpurun_out/quick_tests.log; exit 1; }
tail -1 gpurun_out/quick_tests.log
fi
for m in lane dma; do
for wl in cfg3 cfg2; do
  MUSC_MATCH=$m timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-survey-scope --steps 10 > gpurun_out/quick_${wl}_$m.json 2> gpurun_out/quick_${wl}_$m.err || { tail -5 gpurun_out/quick_${wl}_$m.err; exit 1; }
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/quick_${wl}_$m.json') if l.startswith('{')][-1])
p=d['per_step']
print('$m $wl', 'ms/pass %.3f' % d['ms_per_step'], 'k_match/pass %.3f' % p['ms_screen'], 'select %.3f' % p['ms_select'], 'scan %.3f' % p['ms_scan'], 'device %.3f' % p['ms_device_total'], 'hits', p['hits'], 'roofline', d['roofline'].get('frac'))
PY
done
done
