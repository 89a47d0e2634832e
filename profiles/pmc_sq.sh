#!/bin/bash
# instruction mix of a kernel: profiles/pmc_sq.sh [tag] [workload] [kernel-name prefix] -- the library is the product, or
# MUSC_LIB_PATH (a variant built with -DMUSC_LANE_DBG=n: the experiment knobs are compile-time); the tag only names the output
export TMPDIR=/tmp
d=${1:-0}; wl=${2:-cfg3}; kn=${3:-k_match}
out=gpurun_out/pmc_sq_$d
rm -rf $out; mkdir -p $out
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 bench.py --workload $wl --no-cpu-baseline --no-survey-scope --steps 1 --warmup 1 > $out/p$i.log 2>&1 || { echo "pass $i failed: $grp"; grep -m1 "rror" $out/p$i.log; }
done
python3 profiles/pmc_summary.py $out | grep -A20 "^$kn" > gpurun_out/pmc_sq_$d.txt
rm -rf $out
cat gpurun_out/pmc_sq_$d.txt
