#!/bin/bash
# tools/collect_profiles.sh — on the GPU box: for every measured workload the bench line, right
# behind it the rocprofv3 kernel-trace stats of the same workload (the box drifts by a few per cent
# over minutes at its power cap, so the two that have to agree are taken back to back), and PMC
# passes for the kernels DESIGN.md quotes counters of.  Everything lands in gpurun_out/$ROUND/ (default r06) (copy
# what should be judged into profiles/ afterwards; the raw traces are deleted, they are large).
# One rocprofv3 run per counter set, no tracing domains mixed with --pmc.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
ROUND=${ROUND:-r06}
OUT=$ROOT/gpurun_out/$ROUND
mkdir -p $OUT
top() { # csv title
python3 - "$1" "$2" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print(sys.argv[2])
print(f"{'Name':78s} {'Calls':>6s} {'AvgNs':>12s} {'MinNs':>10s} {'MaxNs':>10s} {'Pct':>6s}")
for r in rows[:10]:
    if r['Name'].startswith('void at::') or 'elementwise' in r['Name'] or 'distribution' in r['Name']:
        continue
    print(f"{r['Name'][:78]:78s} {r['Calls']:>6s} {float(r['AverageNs']):12.0f} {r['MinNs']:>10s} {r['MaxNs']:>10s} {r['Percentage']:>6s}")
PY
}
# long enough that the clock ramp of the first launches after idle does not weigh on the averages
COMMON="--steps 400 --warmup 100 --no-cpu-baseline --check 0 --pmc 0 --sustain 0 --e2e 0 --ceiling 0 --also 0"
for spec in "ns4096:" "c2:--workload c2" "c1:--workload c1" "box10_std:--boxcar 10" "box6_std:--boxcar 6" "c3:--workload c3" "wbfm:--workload wbfm" "scanner:--workload scanner" "c2_16k:--workload c2_16k" "c4:--workload c4"; do
  tag=${spec%%:*}; args=${spec#*:}
  if [ -n "${ONLY:-}" ] && ! echo " $ONLY " | grep -q " $tag "; then continue; fi
  cd $ROOT
  case $tag in
    ns4096) timeout 600 python bench.py > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err ;;
    box*) timeout 300 python bench.py $args --steps 100 --warmup 50 --no-cpu-baseline --e2e 0 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err ;;
    *) timeout 400 python bench.py $args --steps 100 --warmup 50 --cpu-seconds 6 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err ;;
  esac
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- python3 $ROOT/bench.py $COMMON $args > $OUT/trace_$tag.log 2>&1
  f=$(find $OUT/trace_$tag -name "*kernel_stats.csv" | head -1)
  top "$f" "rocprofv3 --kernel-trace --stats -- python3 bench.py $COMMON $args" > $OUT/kernel_stats_$tag.txt
  rm -rf $OUT/trace_$tag $OUT/trace_$tag.log
done
cd $ROOT
if [ -n "${ONLY:-}" ] && [ -z "${PMC_ONLY:-}" ]; then du -sh $ROOT/gpurun_out; exit 0; fi  # ONLY="wbfm c3": those workloads again, no PMC passes; PMC_ONLY="c3 wbfm": those PMC sets
pmc() { # tag kernels bench-args...
  local tag=$1 kernels=$2; shift 2
  PMC_KERNELS="$kernels" bash tools/prof_pmc.sh ${ROUND}_$tag "$@" > /dev/null 2>&1
  for k in $kernels; do cp $ROOT/gpurun_out/pmc_${ROUND}_$tag/summary_$k.txt $OUT/pmc_${tag}_$k.txt; done
  rm -rf $ROOT/gpurun_out/pmc_${ROUND}_$tag
}
if [ -z "${PMC_ONLY:-}" ] || echo " $PMC_ONLY " | grep -q " ns4096 "; then pmc ns4096 "k_fused"; fi
if [ -z "${PMC_ONLY:-}" ] || echo " $PMC_ONLY " | grep -q " c3 "; then pmc c3 "k_fused k_deemph_spec_arb" --workload c3; fi
if [ -z "${PMC_ONLY:-}" ] || echo " $PMC_ONLY " | grep -q " wbfm "; then pmc wbfm "k_boxcar_scan k_deemph_spec_lpr" --workload wbfm; fi
if [ -z "${PMC_ONLY:-}" ] || echo " $PMC_ONLY " | grep -q " box10 "; then pmc box10 "k_boxcar_scan" --boxcar 10; fi
if [ -z "${PMC_ONLY:-}" ] || echo " $PMC_ONLY " | grep -q " box6 "; then pmc box6 "k_boxcar_scan" --boxcar 6; fi
if [ -z "${PMC_ONLY:-}" ] || echo " $PMC_ONLY " | grep -q " c4 "; then pmc c4 "k_power_scan" --workload c4; fi
du -sh $ROOT/gpurun_out
head -3 $OUT/kernel_stats_*.txt
