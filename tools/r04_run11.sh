#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04h
mkdir -p $OUT
cd $ROOT
SECONDS=0
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default bench wall: $SECONDS s"
python3 -c "
import json
d=json.loads(open('$OUT/bench_default.json').read().strip().splitlines()[-1]); r=d['roofline']
print('default', d['ms_per_step'], r['launch_ms'], r['frac'], r.get('sustained',{}).get('frac'), d['config']['output_apart'], d['config']['output_placement'])
for k,v in d['also'].items(): print('  ',k, v.get('launch_ms'), v.get('ms_per_step'), v.get('frac'), v.get('step_frac'), v.get('output_apart'), v.get('error'))"
SECONDS=0
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err
echo "driver-shaped bench wall: $SECONDS s"
python3 -c "
import json
d=json.loads(open('$OUT/bench_driver.json').read().strip().splitlines()[-1]); r=d['roofline']
print('driver', d['ms_per_step'], r['launch_ms'], r['frac'], r.get('sustained',{}).get('frac'), d['config']['output_apart'], d['config']['output_placement'])"
