#!/bin/bash
# The committed bench lines of a round (run from the repo root through gpurun):
#   profiles/bench_lines.sh r04 [a|b]     (two halves: a gpurun call is limited to 20 minutes)
tag=${1:-r04}
part=${2:-ab}
o=gpurun_out
if [[ $part == *a* ]]; then
python bench.py --steps 20 --warmup 5 > $o/${tag}_cfg3_bench.json 2> $o/${tag}_cfg3_bench.err || echo "cfg3 FAILED"
python bench.py --steps 20 --warmup 5 --index classic --no-cpu-baseline --no-survey-scope > $o/${tag}_cfg3_classic_bench.json 2>/dev/null || echo "cfg3 classic FAILED"
python bench.py --workload cfg2 --steps 50 --warmup 5 > $o/${tag}_cfg2_bench.json 2>/dev/null || echo "cfg2 FAILED"
python bench.py --workload cfg4shard --steps 20 --warmup 5 --no-cpu-baseline > $o/${tag}_cfg4shard_bench.json 2>/dev/null || echo "cfg4shard FAILED"
python bench.py --workload cfg5shard --steps 10 --warmup 3 --no-cpu-baseline > $o/${tag}_cfg5shard_bench.json 2>/dev/null || echo "cfg5shard FAILED"
# the reference's own scale run (tests/bigtest/test.sh): four windows of 20 bases on gendat data -> wide context buckets
python bench.py --workload bigtest --steps 50 --warmup 5 --no-cpu-baseline > $o/${tag}_bigtest_bench.json 2>/dev/null || echo "bigtest FAILED"
# the second fused implementation (k_match_g: LDS-DMA, four waves per SIMD) and the general (not geometry-specialised) instances
MUSC_MATCH=dma python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-survey-scope > $o/${tag}_cfg3_dma_bench.json 2>/dev/null || echo "cfg3 dma FAILED"
MUSC_NO_SPEC=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-survey-scope > $o/${tag}_cfg3_nospec_bench.json 2>/dev/null || echo "cfg3 nospec FAILED"
fi
if [[ $part == *b* ]]; then
# two ranks on ONE GPU over gloo (RCCL refuses two ranks on one device): the N>1 control flow only
MUSC_BENCH_BACKEND=gloo MUSC_BENCH_DEVICE=0 MUSC_INDEX=classic timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 1 --reads 5000000 > $o/${tag}_n2_rehearsal.json 2> $o/${tag}_n2_rehearsal.err || { echo "n2 FAILED"; tail -5 $o/${tag}_n2_rehearsal.err; }
# one rank in an RCCL group: the N>1 code with device buffers (the transfers have no peer)
MUSC_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --workload cfg4shard --steps 10 --warmup 2 > $o/${tag}_rccl1_rehearsal.json 2> $o/${tag}_rccl1_rehearsal.err || { echo "rccl1 FAILED"; tail -5 $o/${tag}_rccl1_rehearsal.err; }
MUSC_GRAPH=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-survey-scope > $o/${tag}_cfg3_graph_bench.json 2>/dev/null || echo "cfg3 graph FAILED"
# reads with X (0.1 % of the read bases; the database stays X-free): context buckets (k_match_t, RX) and the two-kernel path
python bench.py --xrate 0.001 --x-reads-only --no-cpu-baseline --no-survey-scope --steps 10 > $o/${tag}_cfg3_xreads_bench.json 2>/dev/null || echo "cfg3 xreads FAILED"
MUSC_NO_X_CONTEXT=1 python bench.py --xrate 0.001 --x-reads-only --no-cpu-baseline --no-survey-scope --steps 10 > $o/${tag}_cfg3_xreads_classic_bench.json 2>/dev/null || echo "cfg3 xreads classic FAILED"
# a database with X (0.1 % of its bases; the reads sampled over one get a random base there): context buckets (k_match_t, XM = 2) and the two-kernel path
python bench.py --xrate 0.001 --x-db-only --no-cpu-baseline --no-survey-scope --steps 10 > $o/${tag}_cfg3_xdb_bench.json 2>/dev/null || echo "cfg3 xdb FAILED"
MUSC_NO_X_CONTEXT=1 python bench.py --xrate 0.001 --x-db-only --no-cpu-baseline --no-survey-scope --steps 10 > $o/${tag}_cfg3_xdb_classic_bench.json 2>/dev/null || echo "cfg3 xdb classic FAILED"
# runs beyond 120 bases of context: wide context buckets, and the two-kernel path they took before
for wl in cfg3w3 cfg3r150; do
  python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-survey-scope > $o/${tag}_${wl}_bench.json 2>/dev/null || echo "$wl FAILED"
  MUSC_CONTEXT=narrow python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-survey-scope > $o/${tag}_${wl}_classic_bench.json 2>/dev/null || echo "$wl classic FAILED"
done
fi
for f in cfg3 cfg3_dma cfg3_nospec cfg3_classic cfg2 cfg4shard cfg5shard bigtest cfg3w3 cfg3w3_classic cfg3r150 cfg3r150_classic n2_rehearsal; do
  python - <<PY
import json
try:
    d = json.loads([l for l in open("$o/${tag}_${f}_bench.json" if "$f" != "n2_rehearsal" else "$o/${tag}_n2_rehearsal.json") if l.startswith("{")][-1])
    r = d["roofline"]
    print("%-14s %-48s ms/step %.3f  value %.3g  kernel %s frac %.3f  hits %d" % ("$f", d["config"]["workload"][:48], d["ms_per_step"], d["value"] or 0, r["kernel"][:9], r["frac"], d["per_step"]["hits"]))
except Exception as e:
    print("$f", "unreadable", e)
PY
done
