# round-4 profile set (run on the GPU box through gpurun; results under gpurun_out/r4prof, copied to profiles/ afterwards)
set -u
O=gpurun_out/r4prof
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bench() { # name args...
  n=$1; shift
  timeout 900 python bench.py "$@" > $O/r4_bench_$n.json 2> $O/r4_bench_$n.log
  grep -h "stages" $O/r4_bench_$n.log | tail -1 | cut -c1-240
}
stats() { # name args...   rocprofv3 --kernel-trace --stats of the bench command + summary of the trace database
  n=$1; shift
  rm -rf /tmp/kt_$n; mkdir -p /tmp/kt_$n
  (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats -d /tmp/kt_$n -o k --output-format csv -- python $R/bench.py "$@" > /tmp/kt_$n/log 2>&1)
  f=$(find /tmp/kt_$n -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/r4_${n}_rocprofv3_kernel_stats.csv
  python tools/kernel_trace.py --sum -- python bench.py "$@" > $O/r4_${n}_kernel_stats.txt 2>&1
}
pmc() { # name args...
  n=$1; shift
  timeout 1800 python tools/pmc_hbm.py $O/r4_${n}_pmc_hbm.json -- python bench.py "$@" > $O/r4_${n}_pmc.log 2>&1
}
W="--no-cpu-baseline --companion 0"
case "${1:-all}" in
bench)
  bench cfg3 --steps 10 --warmup 2
  bench cfg2 --workload cfg2 --steps 10 --warmup 2 --no-cpu-baseline
  bench cfg4_lu --workload cfg4 --steps 5 --warmup 2
  bench cfg4_cg --workload cfg4 --solver cg --rtol 1e-10 --steps 5 --warmup 2
  bench cfg4_cg_chebyshev8 --workload cfg4 --solver cg --pc chebyshev --cheb-degree 8 --rtol 1e-10 --steps 5 --warmup 2
  bench cfg4_cg_chebyshev16 --workload cfg4 --solver cg --pc chebyshev --cheb-degree 16 --rtol 1e-10 --steps 5 --warmup 2
  bench cfg5 --workload cfg5 --rtol 1e-10 --steps 10 --warmup 2
  bench cfg5_bicgstab --workload cfg5 --solver bicgstab --rtol 1e-10 --steps 10 --warmup 2
  TIGAR_PTAP_TENSOR=0 bench cfg3_general_line --steps 3 --warmup 1 $W
  TIGAR_PTAP_TENSOR=0 TIGAR_PTAP_FACTORED=0 TIGAR_IMPLICIT_M=1 bench cfg3_general_hash --steps 2 --warmup 1 $W
  TIGAR_COMM=ipc TIGAR_DEVICE=0 bench cfg2_2ranks_ipc_one_gpu --workload cfg2 --gpus 2 --steps 5 --warmup 2 $W
  TIGAR_COMM=ipc TIGAR_DEVICE=0 bench cfg2_8ranks_ipc_one_gpu --workload cfg2 --gpus 8 --steps 5 --warmup 2 $W
  ;;
general)
  # the general PtAP kernels: workgroup-per-row (tg_ptap.hip) against wave-per-row Gustavson (tg_ptap_wave.hip)
  for w in 1 0; do
    for a in "3 48" "3 64" "3 96" "2 128"; do
      echo "== TIGAR_PTAP_WAVE=$w tools/ptap_bench.py $a"; TIGAR_PTAP_WAVE=$w python tools/ptap_bench.py $a 2>&1 | grep ptap
    done
  done > $O/r4_general_ptap_tensor_operands.txt 2>&1
  for w in 1 0; do for n in 256 512; do TIGAR_PTAP_WAVE=$w python tools/general_ptap_bench.py $n 5; done; done > $O/r4_general_ptap_nonkronecker.jsonl 2>&1
  rm -rf /tmp/kt_nk; mkdir -p /tmp/kt_nk
  (cd /tmp && TIGAR_PTAP_WAVE=0 rocprofv3 --kernel-trace --stats -d /tmp/kt_nk -o k --output-format csv -- python $R/tools/general_ptap_bench.py 512 5 > /tmp/kt_nk/log 2>&1)
  f=$(find /tmp/kt_nk -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r4_general_ptap_nonkronecker_rocprofv3_kernel_stats.csv
  TIGAR_PTAP_WAVE=0 timeout 600 python tools/pmc_hbm.py $O/r4_general_ptap_nonkronecker_pmc_hbm.json -- python tools/general_ptap_bench.py 512 3 > $O/r4_general_ptap_nonkronecker_pmc.log 2>&1
  TIGAR_PTAP_WAVE=1 python tools/pmc_sq.py k_gw -- python tools/ptap_bench.py 3 64 > $O/r4_general_ptap_wave_sq_counters.txt 2>&1
  ;;
stats)
  stats cfg3 --steps 5 --warmup 1 $W
  stats cfg2 --workload cfg2 --steps 5 --warmup 1 $W
  stats cfg4 --workload cfg4 --solver cg --rtol 1e-10 --steps 3 --warmup 1 $W
  stats cfg4_cheb --workload cfg4 --solver cg --pc chebyshev --rtol 1e-10 --steps 3 --warmup 1 $W
  stats cfg5 --workload cfg5 --rtol 1e-10 --steps 3 --warmup 1 $W
  TIGAR_EXTRACT_KRON=0 stats cfg2_general_extraction --workload cfg2 --steps 5 --warmup 1 $W
  TIGAR_EXTRACT_KRON=0 TIGAR_EXTRACT_SEPARABLE=0 stats cfg2_general_extraction_count_fill --workload cfg2 --steps 5 --warmup 1 $W
  ;;
pmc)
  pmc cfg3 --steps 2 --warmup 1 $W
  pmc cfg2 --workload cfg2 --steps 3 --warmup 1 $W
  pmc cfg4 --workload cfg4 --solver cg --rtol 1e-6 --steps 3 --warmup 1 $W
  pmc cfg5 --workload cfg5 --rtol 1e-10 --steps 3 --warmup 1 $W
  TIGAR_EXTRACT_KRON=0 pmc cfg2_general_extraction --workload cfg2 --steps 3 --warmup 1 $W
  TIGAR_EXTRACT_KRON=0 TIGAR_EXTRACT_SEPARABLE=0 pmc cfg2_general_extraction_count_fill --workload cfg2 --steps 3 --warmup 1 $W
  ;;
small)
  # the 2-D configurations again after the persistent Krylov kernels (tg_krylov_small.hip) went in
  stats cfg4 --workload cfg4 --solver cg --rtol 1e-10 --steps 3 --warmup 1 $W
  stats cfg4_cheb --workload cfg4 --solver cg --pc chebyshev --rtol 1e-10 --steps 3 --warmup 1 $W
  stats cfg5 --workload cfg5 --rtol 1e-10 --steps 3 --warmup 1 $W
  pmc cfg4 --workload cfg4 --solver cg --rtol 1e-6 --steps 3 --warmup 1 $W
  pmc cfg5 --workload cfg5 --rtol 1e-10 --steps 3 --warmup 1 $W
  ;;
esac
ls $O | head -80
