O=gpurun_out/r3prof
mkdir -p $O
export TIGAR_EXTRACT_KRON=0
python tools/kernel_trace.py --sum -- python bench.py --workload cfg2 --steps 5 --warmup 1 --no-cpu-baseline --companion 0 > $O/r3_cfg2_general_extraction_kernel_stats.txt 2>&1
timeout 900 python tools/pmc_hbm.py $O/r3_cfg2_general_extraction_pmc_hbm.json -- python bench.py --workload cfg2 --steps 3 --warmup 1 --no-cpu-baseline --companion 0 > $O/r3_cfg2_general_extraction_pmc.log 2>&1
grep -h "extract" $O/r3_cfg2_general_extraction_kernel_stats.txt | head; grep -h "extract" $O/r3_cfg2_general_extraction_pmc.log | head
