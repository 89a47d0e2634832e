#!/bin/bash
# cfg5 shard: the two-kernel path on line buckets against 64-byte buckets
for m in auto classic64; do
  if [ "$m" = "auto" ]; then unset MUSC_INDEX; else export MUSC_INDEX=$m; fi
  timeout -k 10 500 python bench.py --workload cfg5shard --no-cpu-baseline --no-survey-scope --steps 5 > gpurun_out/cfg5_$m.json 2> gpurun_out/cfg5_$m.err || { tail -5 gpurun_out/cfg5_$m.err; continue; }
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/cfg5_$m.json') if l.startswith('{')][-1])
p=d['per_step']
print('$m', 'ms/pass %.3f' % d['ms_per_step'], 'screen %.3f' % p['ms_screen'], 'confirm %.3f' % p['ms_confirm'], 'select %.3f' % p['ms_select'], 'hits', p['hits'], 'desc', p['descriptors'], 'index GB %.1f' % (d['index']['bytes']/1e9), 'build s %.1f' % d['one_off']['index_build_wall_s'])
PY
done
