mkdir -p gpurun_out/r3b
run() { # name, env...
  name=$1; shift
  env "$@" TIGAR_DEVICE=0 timeout 300 python bench.py --workload cfg2 --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --companion 0 > gpurun_out/r3b/$name.json 2> gpurun_out/r3b/$name.log
  echo $name; grep "stages" gpurun_out/r3b/$name.log | cut -c1-200
}
run ipc TIGAR_COMM=ipc
run ipc_noverlap TIGAR_COMM=ipc TIGAR_CG_OVERLAP=0
run ipc_noprio TIGAR_COMM=ipc TIGAR_XSTREAM_PRIO=0
run ipc_look0 TIGAR_COMM=ipc TIGAR_CG_LOOK=0
run host TIGAR_COMM=host
run host_noverlap TIGAR_COMM=host TIGAR_CG_OVERLAP=0
