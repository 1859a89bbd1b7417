# round-5 profile set (run on the GPU box through gpurun; results under gpurun_out/r5prof, copied to profiles/ afterwards)
set -u
O=gpurun_out/r5prof
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W="--no-cpu-baseline --companion 0 --mapped-companion 0 --live-traffic 0"
stats() { # name cmd...   rocprofv3 --kernel-trace --stats of the command + per-kernel summary of the trace database
  n=$1; shift
  rm -rf /tmp/kt_$n; mkdir -p /tmp/kt_$n
  (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d /tmp/kt_$n -o k --output-format csv -- "$@" > /tmp/kt_$n/log 2>&1)
  f=$(find /tmp/kt_$n -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/r5_${n}_rocprofv3_kernel_stats.csv
  python tools/kernel_trace.py --sum -- "$@" > $O/r5_${n}_kernel_stats.txt 2>&1
}
pmc() { n=$1; shift; timeout 1800 python tools/pmc_hbm.py $O/r5_${n}_pmc_hbm.json -- "$@" > $O/r5_${n}_pmc.log 2>&1; }
case "${1:-all}" in
headline)
  timeout 900 python bench.py --steps 10 --warmup 2 > $O/r5_bench_cfg3.json 2> $O/r5_bench_cfg3.log
  stats cfg3 python $R/bench.py --steps 5 --warmup 1 $W
  pmc cfg3 python $R/bench.py --steps 2 --warmup 1 $W
  ;;
mappedbench)
  timeout 900 python bench.py --geometry volume --steps 3 --warmup 1 --no-cpu-baseline --companion 0 --live-traffic 0 > $O/r5_bench_cfg3_mapped_geometry.json 2> $O/r5_bench_cfg3_mapped_geometry.log
  stats cfg3_mapped python $R/bench.py --geometry volume --steps 2 --warmup 1 $W
  pmc cfg3_mapped python $R/bench.py --geometry volume --steps 1 --warmup 1 $W
  ;;
mapped)
  timeout 900 python bench.py --geometry volume --steps 3 --warmup 1 --no-cpu-baseline --companion 0 --live-traffic 0 > $O/r5_bench_cfg3_mapped_geometry.json 2> $O/r5_bench_cfg3_mapped_geometry.log
  stats cfg3_mapped python $R/bench.py --geometry volume --steps 2 --warmup 1 $W
  pmc cfg3_mapped python $R/bench.py --geometry volume --steps 1 --warmup 1 $W
  python tools/pmc_sq.py k_asf3 -- python tools/asm_bench.py 3 64 laplace 1 > $O/r5_assembly_p3_laplace_sq_counters.txt 2>&1
  python tools/pmc_sq.py k_asf3 -- python tools/asm_bench.py 3 64 mass 1 > $O/r5_assembly_p3_mass_sq_counters.txt 2>&1
  for a in "3 64 laplace" "3 64 mass" "2 64 laplace" "2 64 mass" "1 64 laplace" "3 128 laplace"; do TIGAR_ASM_TIME=1 python tools/asm_bench.py $a 3 2>&1 | tail -2; done > $O/r5_assembly_kernels.txt 2>&1
  ;;
rt)
  stats rt64_k1_streamed env TIGAR_IMPLICIT_M=1 python $R/tools/rt_bench.py 64 1 3
  pmc rt64_k1_streamed env TIGAR_IMPLICIT_M=1 python $R/tools/rt_bench.py 64 1 2
  ;;
table)
  stats cfg2 python $R/bench.py --workload cfg2 --steps 10 --warmup 2 $W
  pmc cfg2 python $R/bench.py --workload cfg2 --steps 5 --warmup 1 $W
  stats cfg4 python $R/bench.py --workload cfg4 --solver cg --rtol 1e-10 --steps 3 --warmup 1 $W
  pmc cfg4 python $R/bench.py --workload cfg4 --solver cg --rtol 1e-10 --steps 2 --warmup 1 $W
  stats cfg5 python $R/bench.py --workload cfg5 --rtol 1e-10 --steps 5 --warmup 1 $W
  pmc cfg5 python $R/bench.py --workload cfg5 --rtol 1e-10 --steps 3 --warmup 1 $W
  stats elemsplit python $R/tools/elem_ptap_bench.py 3 64 /dev/null 0
  pmc elemsplit python $R/tools/elem_ptap_bench.py 3 64 /dev/null 0
  ;;
small)
  timeout 600 python bench.py --workload cfg2 --steps 10 --warmup 2 --no-cpu-baseline > $O/r5_bench_cfg2.json 2> $O/r5_bench_cfg2.log
  timeout 600 python bench.py --workload cfg4 --steps 5 --warmup 2 > $O/r5_bench_cfg4_lu.json 2> $O/r5_bench_cfg4_lu.log
  timeout 600 python bench.py --workload cfg4 --solver cg --rtol 1e-10 --steps 5 --warmup 2 > $O/r5_bench_cfg4_cg.json 2> $O/r5_bench_cfg4_cg.log
  timeout 600 python bench.py --workload cfg5 --rtol 1e-10 --steps 10 --warmup 2 > $O/r5_bench_cfg5.json 2> $O/r5_bench_cfg5.log
  TIGAR_COMM=ipc TIGAR_DEVICE=0 timeout 900 python bench.py --workload cfg2 --gpus 2 --steps 5 --warmup 2 $W > $O/r5_bench_cfg2_2ranks_ipc_one_gpu.json 2> $O/r5_bench_cfg2_2ranks.log
  ;;
esac
ls $O | head -60
