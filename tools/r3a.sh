set -x
mkdir -p gpurun_out/r3a
for w in cfg2; do
  timeout 300 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3a/${w}_n1.json 2> gpurun_out/r3a/${w}_n1.log
  TIGAR_COMM=ipc TIGAR_DEVICE=0 timeout 300 python bench.py --workload $w --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3a/${w}_n2_ipc.json 2> gpurun_out/r3a/${w}_n2_ipc.log
  TIGAR_COMM=host TIGAR_DEVICE=0 timeout 300 python bench.py --workload $w --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3a/${w}_n2_host.json 2> gpurun_out/r3a/${w}_n2_host.log
done
timeout 300 python bench.py --workload cfg4 --steps 5 --warmup 2 > gpurun_out/r3a/cfg4_lu.json 2> gpurun_out/r3a/cfg4_lu.log
timeout 300 python bench.py --workload cfg4 --solver cg --rtol 1e-10 --steps 5 --warmup 2 > gpurun_out/r3a/cfg4_cg.json 2> gpurun_out/r3a/cfg4_cg.log
timeout 300 python bench.py --workload cfg5 --rtol 1e-10 --steps 5 --warmup 2 > gpurun_out/r3a/cfg5.json 2> gpurun_out/r3a/cfg5.log
tail -3 gpurun_out/r3a/*.log
