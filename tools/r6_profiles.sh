# round-6 profile set (run on the GPU box through gpurun; results under gpurun_out/r6prof, copied to profiles/ afterwards)
set -u
O=gpurun_out/r6prof
mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W="--no-cpu-baseline --companion 0 --mapped-companion 0 --live-traffic 0"
stats() { # name cmd...   rocprofv3 --kernel-trace --stats of the command + per-kernel summary of the trace database
  n=$1; shift
  rm -rf /tmp/kt_$n; mkdir -p /tmp/kt_$n
  (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d /tmp/kt_$n -o k --output-format csv -- "$@" > /tmp/kt_$n/log 2>&1)
  f=$(find /tmp/kt_$n -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/r6_${n}_rocprofv3_kernel_stats.csv
  python tools/kernel_trace.py --sum -- "$@" > $O/r6_${n}_kernel_stats.txt 2>&1
}
pmc() { n=$1; shift; timeout 1800 python tools/pmc_hbm.py $O/r6_${n}_pmc_hbm.json -- "$@" > $O/r6_${n}_pmc.log 2>&1; }
GEN="env TIGAR_PTAP_TENSOR=0 TIGAR_PTAP_FACTORED=0 TIGAR_IMPLICIT_M=1"
case "${1:-all}" in
elemsplit)
  # the general PtAP by element split, resident 64^3 p = 3 (CSR operands; nothing Kronecker used), against the row-wise kernels
  python tools/elem_ptap_bench.py 3 64 $O/r6_element_split_ptap_64cubed_p3.json 1 > $O/r6_element_split_ptap_64cubed_p3.log 2>&1
  python tools/elem_ptap_bench.py 3 32 $O/r6_element_split_ptap_32cubed_p3.json 1 > /dev/null 2>&1
  python tools/elem_ptap_bench.py 2 96 $O/r6_element_split_ptap_96cubed_p2.json 1 > /dev/null 2>&1
  stats elemsplit python $R/tools/elem_ptap_bench.py 3 64 /dev/null 0
  pmc elemsplit python $R/tools/elem_ptap_bench.py 3 64 /dev/null 0
  python tools/pmc_sq.py k_el_ -- python tools/elem_ptap_bench.py 3 64 /dev/null 0 > $O/r6_elemsplit_sq_counters.txt 2>&1
  ;;
general)
  # cfg3's size with NOTHING assumed about M or A: M materialised chunk by chunk, A in row blocks, element chunks
  $GEN timeout 900 python bench.py --workload cfg3 --steps 3 --warmup 1 $W > $O/r6_bench_cfg3_general_elements.json 2> $O/r6_bench_cfg3_general_elements.log
  stats cfg3_general $GEN python $R/bench.py --workload cfg3 --steps 2 --warmup 1 $W
  pmc cfg3_general $GEN python $R/bench.py --workload cfg3 --steps 1 --warmup 1 $W
  ;;
mapped)
  # element kernels of the mapped forms added in round 6 (a block of the elasticity form; the biharmonic form on the plain kernel)
  stats asm_elast_p3 python $R/tools/asm_bench.py 3 64 elast12 3
  pmc asm_elast_p3 python $R/tools/asm_bench.py 3 64 elast12 3
  python tools/pmc_sq.py k_asf3 -- python tools/asm_bench.py 3 64 elast12 3 > $O/r6_asm_elast_p3_sq_counters.txt 2>&1
  stats asm_elast_p2 python $R/tools/asm_bench.py 2 64 elast12 3
  ;;
quad)
  # p = 3 stiffness / elasticity element kernel, four waves on four consecutive elements, looping over a piece of a line
  # (k_asf3_quad) against the element-per-wave kernel it replaces (TIGAR_ASM_QUAD=0)
  { for q in 0 1; do for a in "3 64 laplace" "3 64 elast12" "3 128 laplace"; do echo "# TIGAR_ASM_QUAD=$q asm_bench $a"; TIGAR_ASM_QUAD=$q TIGAR_ASM_TIME=1 python tools/asm_bench.py $a 3 2>&1 | grep -E "element kernels|ns/element" | tail -2; done; done; } > $O/r6_asm_quad_kernel_times.txt 2>&1
  stats asm_quad_p3 python $R/tools/asm_bench.py 3 64 laplace 3
  pmc asm_quad_p3 python $R/tools/asm_bench.py 3 64 laplace 3
  python tools/pmc_sq.py k_asf3 -- python tools/asm_bench.py 3 64 laplace 3 > $O/r6_asm_quad_p3_sq_counters.txt 2>&1
  TIGAR_ASM_QUAD=0 timeout 600 python tools/pmc_hbm.py $O/r6_asm_elem_p3_pmc_hbm.json -- python tools/asm_bench.py 3 64 laplace 3 > $O/r6_asm_elem_p3_pmc.log 2>&1
  timeout 900 python bench.py --steps 3 --warmup 1 $W --geometry volume > $O/r6_bench_cfg3_mapped_geometry.json 2> $O/r6_bench_cfg3_mapped_geometry.log
  stats cfg3_mapped python $R/bench.py --steps 2 --warmup 1 $W --geometry volume
  pmc cfg3_mapped python $R/bench.py --steps 1 --warmup 1 $W --geometry volume
  ;;
halfstorage)
  # the half-storage products added in the second half of the round: stencil radius 4 (3-D quartics), several fields (elasticity)
  stats symgrid_p4 python $R/tools/symgrid_bench.py 160 4
  pmc symgrid_p4 python $R/tools/symgrid_bench.py 160 4
  stats symgrid_fields env TIGAR_IMPLICIT_M=1 python $R/tools/symgrid_fields_bench.py 96 3
  pmc symgrid_fields env TIGAR_IMPLICIT_M=1 python $R/tools/symgrid_fields_bench.py 96 3
  stats cfg4_cholesky python $R/bench.py --workload cfg4 --solver lu --steps 3 --warmup 1 --no-cpu-baseline --companion 0
  pmc cfg4_cholesky python $R/bench.py --workload cfg4 --solver lu --steps 2 --warmup 1 --no-cpu-baseline --companion 0
  ;;
cholesky)
  # the direct solves on the final tree: cfg4 (narrow band: look-ahead, sweeps) and a 3-D system (wide band: groups of panels)
  timeout 300 python bench.py --workload cfg4 --solver lu --steps 5 --warmup 1 --no-cpu-baseline --companion 0 > $O/r6_bench_cfg4_cholesky.json 2> $O/r6_bench_cfg4_cholesky.log
  stats cfg4_cholesky python $R/bench.py --workload cfg4 --solver lu --steps 3 --warmup 1 --no-cpu-baseline --companion 0
  pmc cfg4_cholesky python $R/bench.py --workload cfg4 --solver lu --steps 2 --warmup 1 --no-cpu-baseline --companion 0
  python tools/kernel_trace.py --sum -- python tools/direct3d_bench.py 2 64 > $O/r6_direct3d_64_kernel_stats.txt 2>&1
  { for a in "2 48" "2 64" "3 40"; do TIGAR_TRACE=1 timeout 300 python tools/direct3d_bench.py $a 2>&1 | grep "default solver\|K:\|workgroups"; done; } > $O/r6_direct3d_final.txt 2>&1
  ;;
headline)
  timeout 900 python bench.py --steps 10 --warmup 2 > $O/r6_bench_cfg3.json 2> $O/r6_bench_cfg3.log
  stats cfg3 python $R/bench.py --steps 5 --warmup 1 $W
  pmc cfg3 python $R/bench.py --steps 2 --warmup 1 $W
  ;;
esac
ls $O | head -60
