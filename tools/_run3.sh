python -m pytest tests/test_parity_gpu.py -q -x -k "c1_boxcar10_fast" 2>&1 | tail -40
