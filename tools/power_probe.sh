RTLFM_PASS0=mfma python bench.py --steps 60000 --warmup 5 --no-cpu-baseline --check 0 > gpurun_out/pw_bench.log 2>&1 &
BP=$!
sleep 30
for i in 1 2 3 4; do rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -iE "power|sclk|mclk|fclk|Temperature" | head -12; echo ---; sleep 1; done
wait $BP
tail -1 gpurun_out/pw_bench.log | cut -c1-300
