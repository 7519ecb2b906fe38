#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04h
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_tail_long_runs_gpu.py tests/test_fullsize_gpu.py tests/test_cli_gpu.py -m gpu -q -k "lpr or tail or wbfm or cli" > $OUT/gpu_tests.txt 2>&1
grep -E "passed|failed|FAILED" $OUT/gpu_tests.txt | tail -6
B="--steps 300 --warmup 100 --no-cpu-baseline --pmc 0 --e2e 0 --check 0 --ceiling 0 --also 0 --sustain 0"
for rnd in 1 2; do
  for mode in "" "lpr_chunk=2048" "lpr_chunk=4096"; do
    export RTLFM_OPTIONS="$mode"
    timeout 300 python bench.py --workload wbfm $B > $OUT/b.json 2> $OUT/b.err
    python3 -c "
import json,sys
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); r=d['roofline']
print('wbfm [$mode]', 'ms/step', d['ms_per_step'], 'launch_ms', r['launch_ms'], 'step_frac', r.get('step_frac'))"
  done
done
unset RTLFM_OPTIONS
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --workload wbfm $B > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:2]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs']):12.0f} {r['MinNs']:>10s} {r['MaxNs']:>10s}")
PY
rm -rf $OUT/trace
cd $ROOT
/usr/bin/time -v python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
grep -E "Elapsed|Maximum resident" $OUT/bench_default.err
python3 -c "
import json
d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print('default', d['ms_per_step'], r['launch_ms'], r['frac'], d['config']['output_apart'], d['config']['output_placement'])
for k,v in d['also'].items(): print('  ',k, v.get('launch_ms'), v.get('ms_per_step'), v.get('frac'), v.get('step_frac'), v.get('error'))"
