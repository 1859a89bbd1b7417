O=gpurun_out/r3prof
mkdir -p $O
( time timeout 1200 python bench.py --steps 10 --warmup 2 > $O/r3_bench_cfg3.json 2> $O/r3_bench_cfg3.log ) 2>&1 | grep real
grep -h "stages\|cpu baseline" $O/r3_bench_cfg3.log | cut -c1-240
TIGAR_COMM=ipc TIGAR_DEVICE=0 timeout 600 python bench.py --workload cfg2 --gpus 8 --steps 5 --warmup 2 --no-cpu-baseline --companion 0 > $O/r3_bench_cfg2_8ranks_ipc_one_gpu.json 2> $O/r3_bench_cfg2_8ranks_ipc_one_gpu.log
grep -h "stages\|self-check" $O/r3_bench_cfg2_8ranks_ipc_one_gpu.log | cut -c1-240
grep MemAvailable /proc/meminfo; nproc
