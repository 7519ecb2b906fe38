sed -i 's/pat = sys.argv\[2\] if len(sys.argv) > 2 else "k_fused"/pat = sys.argv[2] if len(sys.argv) > 2 else os.environ.get("PMC_KERNEL", "k_fused")/' tools/pmc_summary.py
export PMC_KERNEL=k_boxcar
bash tools/prof_pmc.sh box10s --boxcar 10 > /dev/null 2>&1
bash tools/prof_pmc.sh box6s --boxcar 6 > /dev/null 2>&1
cat gpurun_out/pmc_box10s/summary.txt; echo; cat gpurun_out/pmc_box6s/summary.txt
