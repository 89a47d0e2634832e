#!/bin/bash
# cfg5 shard screen/confirm times of library variants: profiles/cfg5v.sh product build_variants/x.so ...
for lib in "$@"; do
  if [ "$lib" = "product" ]; then unset MUSC_LIB_PATH; else export MUSC_LIB_PATH=$PWD/$lib; fi
  timeout -k 10 500 python bench.py --workload cfg5shard --no-cpu-baseline --no-survey-scope --steps 5 > gpurun_out/cfg5v.json 2> gpurun_out/cfg5v.err || { tail -5 gpurun_out/cfg5v.err; continue; }
  python - <<PY
import json
d=json.loads([l for l in open('gpurun_out/cfg5v.json') if l.startswith('{')][-1])
p=d['per_step']
print('$lib', 'ms/pass %.3f' % d['ms_per_step'], 'screen %.3f' % p['ms_screen'], 'confirm %.3f' % p['ms_confirm'], 'hits', p['hits'], 'desc', p['descriptors'])
PY
done
