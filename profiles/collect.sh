#!/bin/bash
# Collect the round's profile evidence on the GPU box (run from the repo root through gpurun):
#   profiles/collect.sh r01
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/ and the summaries next to it;
# copy <tag>_* into profiles/ afterwards.  Counters are collected in their own passes
# (never together with --kernel-trace/--stats), one counter group per pass.
set -o pipefail
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
B="python3 bench.py --workload cfg3 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $B --steps 5 > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 1; }
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_REQ_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  # (a counter group the hardware cannot collect in one pass aborts the run: bounded, and skipped)
  if timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $out/pmc$i -- $B --steps 1 --warmup 1 > $out/pmc$i.log 2>&1; then
    echo "pmc pass $i done: $grp"
  else
    echo "pmc pass $i FAILED: $grp"; grep -m1 "failed with error" $out/pmc$i.log
  fi
done
f=$(find $out/stats -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/${tag}_cfg3_kernel_stats.csv
python3 profiles/pmc_summary.py $out > gpurun_out/${tag}_cfg3_pmc_summary.txt
grep "^{" $out/stats.log > gpurun_out/${tag}_cfg3_bench_under_rocprof.json
head -12 gpurun_out/${tag}_cfg3_kernel_stats.csv
