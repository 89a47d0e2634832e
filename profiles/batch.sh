#!/bin/bash
# cfg3 pass time against the batch size (MUSC_BATCH_READS)
for b in 16777216 8388608 5592406 4194304; do
  MUSC_BATCH_READS=$b timeout -k 10 200 python bench.py --workload cfg3 --no-cpu-baseline --no-survey-scope --steps 20 > gpurun_out/bt.json 2> gpurun_out/bt.err
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/bt.json') if l.startswith('{')][-1])
p=d['per_step']
print('batch $b', 'ms/pass %.3f' % d['ms_per_step'], 'k_match/pass %.3f' % p['ms_screen'], 'select %.3f' % p['ms_select'], 'scan %.3f' % p['ms_scan'], 'device %.3f' % p['ms_device_total'], 'first %.3f' % d.get('first_pass_ms', 0))
PY
done
