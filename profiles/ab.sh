#!/bin/bash
# A/B on one box: the quad kernel (MUSC_MATCH=quad) against the dense one, cfg3, 20 passes each, twice
for rep in 1 2; do
for m in quad dense; do
  MUSC_MATCH=$m timeout -k 10 200 python bench.py --workload cfg3 --no-cpu-baseline --no-survey-scope --steps 20 > gpurun_out/ab.json 2> gpurun_out/ab.err
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/ab.json') if l.startswith('{')][-1])
p=d['per_step']
print('$m', 'ms/pass %.3f' % d['ms_per_step'], 'k_match/launch %.4f' % (p['ms_screen']/3), 'select %.3f' % p['ms_select'], 'hits', p['hits'])
PY
done
done
