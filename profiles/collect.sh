#!/bin/bash
# Collect a round's profile evidence on the GPU box (run from the repo root through gpurun):
#   profiles/collect.sh r04 [workload ...]        default workloads: cfg3 cfg2 cfg4shard cfg5shard bigtest cfg3w3 cfg3r150 (also: cfg3xdb cfg3xreads cfg3xdb_classic cfg3xreads_classic cfg3w3_classic cfg3r150_classic)
# Per workload: rocprofv3 --kernel-trace --stats of `bench.py --workload W --steps 5` (kernel
# statistics + the JSON line of that very run), then separate --pmc passes (never combined with a
# trace; one counter group per pass; the program directly after `--`) for HBM traffic.  cfg3 also gets
# the two-kernel path (--index classic) and the second fused kernel (MUSC_MATCH=dma: k_match_g); the SQ
# instruction mix is profiles/pmc_sq.sh.  Raw output stays under
# gpurun_out/prof_<tag>/; the summaries land in gpurun_out/<tag>_* -- copy those into profiles/.
set -o pipefail
tag=${1:-r04}; shift
wls=${@:-cfg3 cfg2 cfg4shard cfg5shard bigtest cfg3w3 cfg3r150}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
echo "{" > gpurun_out/${tag}_traffic.json.parts
first=1
for wl in $wls; do
  kinds="auto"; [ "$wl" = "cfg3" ] && kinds="auto classic"
  # cfg3 with 0.1 % X: in the database alone (cfg3xdb: context buckets, k_match_t<.., XM = 2>), in the reads alone (cfg3xreads: XM = 1)
  bwl=$wl; xflags=""
  unset MUSC_NO_X_CONTEXT
  case "$wl" in
    cfg3xdb*) bwl=cfg3; xflags="--xrate 0.001 --x-db-only";;
    cfg3xreads*) bwl=cfg3; xflags="--xrate 0.001 --x-reads-only";;
  esac
  # (..._classic: the same run kept off the context buckets -- the two-kernel path it took before)
  case "$wl" in cfg3x*_classic) export MUSC_NO_X_CONTEXT=1;; esac
  # (cfg3w3_classic, cfg3r150_classic: the wide-bucket workloads on the two-kernel path they took before r03)
  unset MUSC_CONTEXT
  case "$wl" in cfg3w3_classic|cfg3r150_classic) bwl=${wl%_classic}; export MUSC_CONTEXT=narrow;; esac
  for kind in $kinds; do
    B="python3 bench.py --workload $bwl $xflags --no-cpu-baseline --no-survey-scope --index $kind"
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_${wl}_$kind -- $B --steps 5 > $out/stats_${wl}_$kind.log 2>&1 || { tail -5 $out/stats_${wl}_$kind.log; exit 1; }
    f=$(find $out/stats_${wl}_$kind -name "*kernel_stats.csv" | head -1)
    grep -E "^\"?Name|k_" "$f" > gpurun_out/${tag}_${wl}_${kind}_kernel_stats.csv
    grep "^{" $out/stats_${wl}_$kind.log > gpurun_out/${tag}_${wl}_${kind}_bench_under_rocprof.json
    i=0
    for grp in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
               "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
               "TCC_EA0_RDREQ_DRAM_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
      i=$((i+1))
      if timeout -k 10 400 rocprofv3 --pmc $grp --output-format csv -d $out/pmc_${wl}_${kind}_$i -- $B --steps 1 --warmup 1 > $out/pmc_${wl}_${kind}_$i.log 2>&1; then
        echo "$wl $kind pmc pass $i done: $grp"
      else
        echo "$wl $kind pmc pass $i FAILED: $grp"; grep -m1 "failed with error" $out/pmc_${wl}_${kind}_$i.log
      fi
    done
    mkdir -p $out/pmcs_${wl}_$kind; mv $out/pmc_${wl}_${kind}_[0-9] $out/pmcs_${wl}_$kind/ 2>/dev/null
    key=$wl; [ "$kind" = "classic" ] && key=${wl}_classic
    python3 profiles/traffic_from_pmc.py $out/pmcs_${wl}_$kind $key k_match_t k_match_g k_screen_t k_screen k_confirm k_compact_w k_compact > $out/traffic_$key.json
    rm -rf $out/pmcs_${wl}_$kind $out/stats_${wl}_$kind/*/*.db 2>/dev/null
  done
done
python3 - "$out" "$tag" <<'PY'
import glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
m = {}
for f in sorted(glob.glob(out + "/traffic_*.json")):
    m.update(json.load(open(f)))
m["method"] = ("rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE ..., --pmc TCC_EA0_RDREQ* in separate passes of `bench.py --workload W "
               "--steps 1 --warmup 1 [--index classic]` (profiles/collect.sh), mean over the kernel's dispatches; FETCH_SIZE "
               "doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at 64 B), cross-checked against the "
               "TCC_EA0_RDREQ counters by request size; memory-side counters include Infinity-Cache hits")
json.dump(m, open("gpurun_out/%s_traffic.json" % tag, "w"), indent=1)
print(json.dumps({k: {kk: round(vv["traffic_bytes_per_launch"] / 1e9, 3) for kk, vv in v.items()} for k, v in m.items() if k != "method"}))
PY
rm -f gpurun_out/${tag}_traffic.json.parts
if echo " $wls " | grep -q " cfg3 "; then
  # k_match_g (the second fused kernel on the same buckets), kernel statistics only
  MUSC_MATCH=dma timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_dma -- python3 bench.py --workload cfg3 --no-cpu-baseline --no-survey-scope --steps 5 > $out/stats_dma.log 2>&1 || tail -3 $out/stats_dma.log
  f=$(find $out/stats_dma -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && grep -E "^\"?Name|k_" "$f" > gpurun_out/${tag}_cfg3_dma_kernel_stats.csv
  # (no copy + kernel time line of the SURVEY-scope leg: rocprofv3 --memory-copy-trace dies at exit on this image, profiles/README.md)
  rm -rf $out/stats_dma/*/*.db 2>/dev/null
fi
du -sh $out
