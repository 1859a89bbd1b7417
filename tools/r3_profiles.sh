# round-3 profile set (run on the GPU box through gpurun; results under gpurun_out/r3prof, copied to profiles/ afterwards)
set -u
O=gpurun_out/r3prof
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bench() { # name args...
  n=$1; shift
  timeout 900 python bench.py "$@" > $O/r3_bench_$n.json 2> $O/r3_bench_$n.log
  grep -h "stages" $O/r3_bench_$n.log | cut -c1-240
}
stats() { # name args...   rocprofv3 --kernel-trace --stats of the bench command + summary of the trace database
  n=$1; shift
  rm -rf /tmp/kt_$n; mkdir -p /tmp/kt_$n
  (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d /tmp/kt_$n -o k --output-format csv -- python $R/bench.py "$@" > /tmp/kt_$n/log 2>&1)
  f=$(find /tmp/kt_$n -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/r3_${n}_rocprofv3_kernel_stats.csv
  python tools/kernel_trace.py --sum -- python bench.py "$@" > $O/r3_${n}_kernel_stats.txt 2>&1
}
pmc() { # name args...
  n=$1; shift
  timeout 1800 python tools/pmc_hbm.py $O/r3_${n}_pmc_hbm.json -- python bench.py "$@" > $O/r3_${n}_pmc.log 2>&1
}
W="--no-cpu-baseline --companion 0"
case "${1:-all}" in
bench)
  bench cfg3 --steps 10 --warmup 2
  bench cfg2 --workload cfg2 --steps 10 --warmup 2 --no-cpu-baseline
  bench cfg4_lu --workload cfg4 --steps 5 --warmup 2
  bench cfg4_cg --workload cfg4 --solver cg --rtol 1e-10 --steps 5 --warmup 2
  bench cfg5 --workload cfg5 --rtol 1e-10 --steps 10 --warmup 2
  TIGAR_PTAP_FUSED=0 bench cfg3_fe_matrix_materialised --steps 5 --warmup 1 $W
  TIGAR_PTAP_FUSED=0 TIGAR_PTAP_VERIFY=1 bench cfg3_pattern_verified --steps 5 --warmup 1 $W
  TIGAR_PTAP_TENSOR=0 bench cfg3_general_line --steps 3 --warmup 1 $W
  TIGAR_PTAP_TENSOR=0 TIGAR_PTAP_FACTORED=0 TIGAR_IMPLICIT_M=1 bench cfg3_general_hash --steps 2 --warmup 1 $W
  TIGAR_COMM=ipc TIGAR_DEVICE=0 bench cfg2_2ranks_ipc_one_gpu --workload cfg2 --gpus 2 --steps 5 --warmup 2 $W
  TIGAR_COMM=ipc TIGAR_DEVICE=0 bench cfg3_8ranks_ipc_one_gpu --gpus 8 --steps 2 --warmup 1 $W
  ;;
stats)
  stats cfg3 --steps 5 --warmup 1 $W
  stats cfg2 --workload cfg2 --steps 5 --warmup 1 $W
  stats cfg4 --workload cfg4 --solver cg --rtol 1e-10 --steps 3 --warmup 1 $W
  stats cfg4_lu --workload cfg4 --solver lu --steps 2 --warmup 1 $W
  stats cfg5 --workload cfg5 --rtol 1e-10 --steps 3 --warmup 1 $W
  ;;
pmc)
  pmc cfg3 --steps 2 --warmup 1 $W
  pmc cfg2 --workload cfg2 --steps 3 --warmup 1 $W
  pmc cfg4 --workload cfg4 --solver cg --rtol 1e-6 --steps 3 --warmup 1 $W
  pmc cfg5 --workload cfg5 --rtol 1e-10 --steps 3 --warmup 1 $W
  ;;
esac
ls $O | head -60
