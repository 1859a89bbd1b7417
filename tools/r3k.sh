O=gpurun_out/r3prof
mkdir -p $O
timeout 600 python bench.py --workload cfg4 --steps 5 --warmup 2 > $O/r3_bench_cfg4_lu.json 2> $O/r3_bench_cfg4_lu.log
grep -h stages $O/r3_bench_cfg4_lu.log | cut -c1-200
bash tools/r3h.sh
