#!/bin/bash
# Collect the round's profile evidence on the GPU box (run from the repo root through gpurun):
#   profiles/collect.sh r02
# Writes raw rocprofv3 output under gpurun_out/prof_<tag>/ and the summaries next to it; copy
# gpurun_out/<tag>_* into profiles/ afterwards.  Counters are collected in their own passes (never
# together with --kernel-trace/--stats), one counter group per pass.  Two index kinds: "auto" = what
# the library picks for cfg3 (context buckets, k_match_d) and "classic" (64-byte buckets, k_screen ->
# k_confirm); the kernel statistics also with MUSC_MATCH=quad (context buckets, k_match).
set -o pipefail
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
for kind in auto classic; do
  B="python3 bench.py --workload cfg3 --no-cpu-baseline --no-survey-scope --index $kind"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$kind -- $B --steps 5 > $out/stats_$kind.log 2>&1 || { tail -5 $out/stats_$kind.log; exit 1; }
  f=$(find $out/stats_$kind -name "*kernel_stats.csv" | head -1)
  grep -E "^\"?Name|k_" "$f" > gpurun_out/${tag}_cfg3_${kind}_kernel_stats.csv
  grep "^{" $out/stats_$kind.log > gpurun_out/${tag}_cfg3_${kind}_bench_under_rocprof.json
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
             "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
             "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    # (a counter group the hardware cannot collect in one pass aborts the run: bounded, and skipped)
    if timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $out/pmc_${kind}_$i -- $B --steps 1 --warmup 1 > $out/pmc_${kind}_$i.log 2>&1; then
      echo "$kind pmc pass $i done: $grp"
    else
      echo "$kind pmc pass $i FAILED: $grp"; grep -m1 "failed with error" $out/pmc_${kind}_$i.log
    fi
  done
done
# k_match (the kernel k_match_d replaced for up to two windows), kernel statistics only
MUSC_MATCH=quad timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_quad -- python3 bench.py --workload cfg3 --no-cpu-baseline --no-survey-scope --steps 5 > $out/stats_quad.log 2>&1 || tail -3 $out/stats_quad.log
f=$(find $out/stats_quad -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -E "^\"?Name|k_" "$f" > gpurun_out/${tag}_cfg3_quad_kernel_stats.csv
# roctx ranges of the library (marker trace is not a counter pass)
timeout -k 10 300 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $out/marker -- python3 bench.py --workload cfg3 --no-cpu-baseline --no-survey-scope --steps 3 > $out/marker.log 2>&1 || tail -3 $out/marker.log
f=$(find $out/marker -name "*marker_api_stats.csv" -o -name "*marker*stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_cfg3_marker_stats.csv
python3 profiles/pmc_summary.py $out > gpurun_out/${tag}_cfg3_pmc_summary.txt
python3 profiles/traffic_from_pmc.py $out cfg3 k_match_d k_match k_screen k_confirm k_compact_w k_compact > gpurun_out/${tag}_traffic.json
rm -rf $out/pmc_* $out/stats_*/*/*.db 2>/dev/null; du -sh $out; cat gpurun_out/${tag}_cfg3_auto_kernel_stats.csv | head -8
cat gpurun_out/${tag}_traffic.json | head -60
