python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py -q -n 4 -k "box or wbfm or c1 or random or usb" 2>&1 | tail -8
for D in 10 6 16; do
python bench.py --boxcar $D --steps 100 --warmup 50 --no-cpu-baseline --check 0 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scan D=$D', d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])"
done
python bench.py --boxcar 6 --atan fast --steps 100 --warmup 50 --no-cpu-baseline --check 0 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scan D=6 fast', d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])"
python bench.py --boxcar 7 --steps 100 --warmup 50 --no-cpu-baseline --check 0 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('scan D=7', d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])"
