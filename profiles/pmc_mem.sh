#!/bin/bash
# memory-pipeline counters of the match kernel (one counter group per pass, no trace): profiles/pmc_mem.sh [tag] [workload]
# the library is the product, or MUSC_LIB_PATH / MUSC_MATCH as the environment says
export TMPDIR=/tmp
d=${1:-0}; wl=${2:-cfg3}
out=gpurun_out/pmc_mem_$d
rm -rf $out; mkdir -p $out
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_BUSY_avr TCC_TAG_STALL_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
           "FETCH_SIZE WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $out/p$i -- python3 bench.py --workload $wl --no-cpu-baseline --no-survey-scope --steps 1 --warmup 1 > $out/p$i.log 2>&1 || { echo "pass $i failed: $grp"; grep -m1 "rror" $out/p$i.log; }
done
python3 profiles/pmc_summary.py $out | grep -A40 "^k_match" | grep -B100 -m2 "^k_" | head -60 > gpurun_out/pmc_mem_$d.txt
rm -rf $out
cat gpurun_out/pmc_mem_$d.txt
