#!/bin/bash
# instruction-cache behaviour of the match kernel (library given by MUSC_LIB_PATH, else the built one)
export TMPDIR=/tmp
tag=${1:-x}
out=gpurun_out/pmc_ic_$tag
rm -rf $out; mkdir -p $out
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVES" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_IFETCH SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 bench.py --workload cfg3 --no-cpu-baseline --no-survey-scope --steps 1 --warmup 1 > $out/p$i.log 2>&1 || { echo "pass $i failed: $grp"; grep -m1 "rror" $out/p$i.log; }
done
python3 profiles/pmc_summary.py $out | grep -A8 "k_match" > gpurun_out/pmc_ic_$tag.txt
rm -rf $out
cat gpurun_out/pmc_ic_$tag.txt
