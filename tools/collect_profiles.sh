#!/bin/bash
# tools/collect_profiles.sh <tag> — on the GPU box: bench line, rocprofv3 kernel-trace stats
# and PMC passes for the default workload; everything lands in gpurun_out/<tag>/ (copy the
# summaries you want judged into profiles/ afterwards).
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --check 0 --pmc 0 --sustain 0 > $OUT/trace.log 2>&1
cd $ROOT
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" > $OUT/kernel_stats_top.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("rocprofv3 --kernel-trace --stats  (python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --check 0 --pmc 0 --sustain 0)")
print(f"{'Name':70s} {'Calls':>6s} {'AvgNs':>12s} {'MinNs':>10s} {'MaxNs':>10s} {'Pct':>6s}")
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs']):12.0f} {r['MinNs']:>10s} {r['MaxNs']:>10s} {r['Percentage']:>6s}")
PY
cat $OUT/kernel_stats_top.txt | grep -E "k_fused|Name"
bash tools/prof_pmc.sh $TAG > $OUT/pmc.log 2>&1
cp $ROOT/gpurun_out/pmc_$TAG/summary.txt $OUT/pmc_summary.txt
cat $OUT/pmc_summary.txt
