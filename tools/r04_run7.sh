#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04g
mkdir -p $OUT
cd $ROOT
B="--steps 300 --warmup 100 --no-cpu-baseline --pmc 0 --e2e 0 --check 0 --ceiling 0 --also 0 --sustain 0"
for rnd in 1 2 3; do
for w in c3 wbfm; do
  for mode in "" "tail_priority=1" "tail_priority=-1" "tail_serial=1" "tail_priority=1,fused_waves_tail=8192" "fused_waves_tail=8192"; do
    export RTLFM_OPTIONS="$mode"
    timeout 300 python bench.py --workload $w $B > $OUT/b.json 2> $OUT/b.err
    python3 -c "
import json,sys
d=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$w [$mode]', 'ms/step', d['ms_per_step'], 'launch_ms', r['launch_ms'], 'step_frac', r.get('step_frac'))" | tee -a $OUT/prio.txt
  done
done
done
