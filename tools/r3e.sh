mkdir -p gpurun_out/r3e
timeout 300 python bench.py --workload cfg4 --steps 5 --warmup 2 > gpurun_out/r3e/cfg4_lu.json 2> gpurun_out/r3e/cfg4_lu.log
timeout 300 python bench.py --workload cfg4 --solver cg --rtol 1e-10 --steps 5 --warmup 2 > gpurun_out/r3e/cfg4_cg.json 2> gpurun_out/r3e/cfg4_cg.log
timeout 300 python bench.py --workload cfg5 --rtol 1e-10 --steps 5 --warmup 2 > gpurun_out/r3e/cfg5.json 2> gpurun_out/r3e/cfg5.log
timeout 300 python bench.py --workload cfg2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3e/cfg2.json 2> gpurun_out/r3e/cfg2.log
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r3e/cfg3.json 2> gpurun_out/r3e/cfg3.log
grep -h "stages" gpurun_out/r3e/*.log | cut -c1-260
