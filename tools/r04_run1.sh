#!/bin/bash
# round 4, first GPU session: the parity suite on the new code, the new default bench line, c3 / wbfm lines + traces
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04a
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.txt 2>&1
tail -5 $OUT/gpu_tests.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/bench_default.err
for w in c3 wbfm; do
  timeout 300 python bench.py --workload $w --steps 300 --warmup 100 --no-cpu-baseline --pmc 0 --e2e 0 --check 0 > $OUT/bench_$w.json 2> $OUT/bench_$w.err
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 $ROOT/bench.py --workload $w --steps 300 --warmup 100 --no-cpu-baseline --check 0 --pmc 0 --sustain 0 --e2e 0 --ceiling 0 --also 0 > $OUT/trace_$w.log 2>&1
  f=$(find $OUT/trace_$w -name "*kernel_stats.csv" | head -1)
  python3 - "$f" > $OUT/kernel_stats_$w.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print(f"{'Name':78s} {'Calls':>6s} {'AvgNs':>12s} {'MinNs':>10s} {'MaxNs':>10s} {'Pct':>6s}")
for r in rows[:14]:
    print(f"{r['Name'][:78]:78s} {r['Calls']:>6s} {float(r['AverageNs']):12.0f} {r['MinNs']:>10s} {r['MaxNs']:>10s} {r['Percentage']:>6s}")
PY
  rm -rf $OUT/trace_$w
  cd $ROOT
done
cat $OUT/kernel_stats_c3.txt
