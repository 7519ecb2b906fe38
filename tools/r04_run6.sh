#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04f
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_tail_long_runs_gpu.py tests/test_fullsize_gpu.py -m gpu -q \
  -k "deemph or lpr or tail or wbfm or c3 or c5 or golden" > $OUT/gpu_tests.txt 2>&1
grep -E "passed|failed|FAILED" $OUT/gpu_tests.txt | tail -12
B="--steps 300 --warmup 100 --no-cpu-baseline --pmc 0 --e2e 0 --check 0 --ceiling 0 --also 0 --sustain 0"
for rnd in 1 2; do
for w in c3 wbfm; do
  for mode in float integer; do
    if [ $w = c3 ] && [ $mode = integer ]; then continue; fi
    if [ $mode = integer ]; then export RTLFM_OPTIONS="deemph_integer=1"; else unset RTLFM_OPTIONS; fi
    timeout 300 python bench.py --workload $w $B > $OUT/bench_${w}_$mode.json 2> $OUT/bench_${w}_$mode.err
    python3 -c "
import json,sys
d=json.loads(open('$OUT/bench_${w}_$mode.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$w $mode', 'ms/step', d['ms_per_step'], 'launch_ms', r['launch_ms'], 'step_frac', r.get('step_frac'))"
  done
  unset RTLFM_OPTIONS
done
done
for w in c3 wbfm; do
  cd /tmp && export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$w -- python3 $ROOT/bench.py --workload $w $B > $OUT/trace_$w.log 2>&1
  f=$(find $OUT/trace_$w -name "*kernel_stats.csv" | head -1)
  python3 - "$f" > $OUT/kernel_stats_$w.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print(f"{'Name':78s} {'Calls':>6s} {'AvgNs':>12s} {'MinNs':>10s} {'MaxNs':>10s} {'Pct':>6s}")
for r in rows[:3]:
    print(f"{r['Name'][:78]:78s} {r['Calls']:>6s} {float(r['AverageNs']):12.0f} {r['MinNs']:>10s} {r['MaxNs']:>10s} {r['Percentage']:>6s}")
PY
  rm -rf $OUT/trace_$w
  cd $ROOT
  cat $OUT/kernel_stats_$w.txt
done
