#!/bin/bash
# the SURVEY-scope leg under a copy + kernel trace (no counters): profiles/copies.sh r03
export TMPDIR=/tmp
tag=${1:-r03}
out=gpurun_out/prof_$tag
mkdir -p $out
for fmt in csv json; do
  rm -rf $out/copies
  timeout -k 10 300 rocprofv3 --memory-copy-trace --kernel-trace --output-format $fmt -d $out/copies -- python3 bench.py --workload cfg3 --no-cpu-baseline --steps 2 > $out/copies_$fmt.log 2>&1
  echo "format $fmt: exit $?"; find $out/copies -type f | head
  if [ -n "$(find $out/copies -name '*memory_copy_trace.csv')" ]; then
    python3 profiles/overlap_from_trace.py $out/copies > gpurun_out/${tag}_cfg3_survey_scope_overlap.txt 2>&1
    cat gpurun_out/${tag}_cfg3_survey_scope_overlap.txt | head -40
    break
  fi
done
