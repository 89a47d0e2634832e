#!/bin/bash
# the round's fuzz totals with the final library (run from the repo root through gpurun): profiles/fuzz_r04.sh [tag]
tag=${1:-r04}
o=gpurun_out/${tag}_fuzz_totals.txt
echo "Round 4, final library (k_match_t incl. the geometry-specialised instance where a case's geometry matches; k_match_g with MUSC_MATCH=dma), one MI355X box:" > $o
run() { echo "--- $*" >> $o; timeout -k 10 900 env "$@" 2>&1 | grep -E "bad|MISMATCH" | tail -5 >> $o; }
run X=0 python tests/fuzz_gpu.py 0 60000
run MUSC_MATCH=dma python tests/fuzz_gpu.py 0 40000
run X=0 python tests/fuzz_gpu_medium.py 0 14000
run MUSC_MATCH=dma python tests/fuzz_gpu_medium.py 14000 26000
run MUSC_FUZZ_READS_X=1 python tests/fuzz_gpu_medium.py 0 5000
run MUSC_FUZZ_DB_X=1 python tests/fuzz_gpu_medium.py 0 4000
run MUSC_FUZZ_DB_X=2 python tests/fuzz_gpu_medium.py 4000 8000
cat $o
