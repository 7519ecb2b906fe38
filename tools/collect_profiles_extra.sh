#!/bin/bash
# tools/collect_profiles_extra.sh <tag> — rocprofv3 kernel-trace stats for the workloads that are
# not the default bench line: fused boxcar /10 (config-1 shaped), C3 (4096 NBFM streams with the
# audio tail) and rtl_power (config 4).  Output: gpurun_out/<tag>/*_top.txt
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
top() { # csv title
python3 - "$1" "$2" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "rtlfm" in r["Name"] or "rtlpower" in r["Name"]]
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print(sys.argv[2])
print(f"{'Name':78s} {'Calls':>6s} {'AvgNs':>12s} {'MinNs':>10s} {'MaxNs':>10s}")
for r in rows[:10]:
    print(f"{r['Name'][:78]:78s} {r['Calls']:>6s} {float(r['AverageNs']):12.0f} {r['MinNs']:>10s} {r['MaxNs']:>10s}")
PY
}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/boxcar -- python3 $ROOT/bench.py --boxcar 10 --steps 20 --warmup 5 --no-cpu-baseline --check 0 > $OUT/boxcar.log 2>&1
top $(find $OUT/boxcar -name "*kernel_stats.csv" | head -1) "rocprofv3 --kernel-trace --stats  (python3 bench.py --boxcar 10 --steps 20 --warmup 5 --no-cpu-baseline --check 0)" > $OUT/kernel_stats_boxcar10_top.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -- python3 $ROOT/tools/bench_c3.py > $OUT/c3.log 2>&1
top $(find $OUT/c3 -name "*kernel_stats.csv" | head -1) "rocprofv3 --kernel-trace --stats  (python3 tools/bench_c3.py: 4096 streams x 4 x 262144 B, /64 + FIR9 + deemph + arbitrary_resample)" > $OUT/kernel_stats_c3_top.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/power -- python3 $ROOT/tools/bench_power.py --steps 5 --cpu-seconds 0 > $OUT/power.log 2>&1
top $(find $OUT/power -name "*kernel_stats.csv" | head -1) "rocprofv3 --kernel-trace --stats  (python3 tools/bench_power.py --steps 5: 1024 streams x 64 reads x 32768 B, 2^14-bin fix_fft)" > $OUT/kernel_stats_power_top.txt
cat $OUT/*_top.txt
