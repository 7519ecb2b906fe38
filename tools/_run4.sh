python -m pytest tests/test_parity_gpu.py -q -n 4 -k "c2 or c3 or golden" 2>&1 | tail -3
RTLFM_HIP_LIB=$PWD/build_ablate/lib_h4.so python -m pytest tests/test_parity_gpu.py -q -n 4 -k "c2 or c3 or golden" 2>&1 | tail -3
for P in 4 5; do
for v in h4 h8; do
echo "P=$P h1 vs $v"
python tools/ab_engines.py --passes $P --paths 0 0 --libs build_ablate/lib_h1.so build_ablate/lib_$v.so --rounds 20 2>&1 | tail -2
done; done
