#!/bin/bash
# further seed ranges with the round's final library (run from the repo root through gpurun): profiles/fuzz_r04b.sh [tag]
tag=${1:-r04}
o=gpurun_out/${tag}_fuzz_totals_b.txt
echo "Round 4, final library, seed ranges beyond profiles/fuzz_r04.sh's, one MI355X box:" > $o
run() { echo "--- $*" >> $o; timeout -k 10 900 env "$@" 2>&1 | grep -E "bad|MISMATCH" | tail -5 >> $o; }
run X=0 python tests/fuzz_gpu.py 60000 160000
run MUSC_MATCH=dma python tests/fuzz_gpu.py 160000 220000
run X=0 python tests/fuzz_gpu_medium.py 26000 40000
run MUSC_FUZZ_READS_X=1 python tests/fuzz_gpu_medium.py 5000 10000
run MUSC_FUZZ_DB_X=1 python tests/fuzz_gpu_medium.py 8000 12000
cat $o
