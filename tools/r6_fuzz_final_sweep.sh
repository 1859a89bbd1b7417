# sweep of round 6 on its tree: fresh seeds, the switch sets of tests/test_gpu_fuzz.py and the new paths (element chunks)
S=${1:-6900}; O=gpurun_out/r6fuzz; mkdir -p $O; F=$O/r6_fuzz_sweep_$S.txt; : > $F
run() { n=$1; shift; echo "== $n: $*" >> $F; ( "$@" 2>&1 | grep "^{\|^FAIL" | tail -4 | cut -c1-700 ) >> $F; }
GEN="TIGAR_PTAP_TENSOR=0 TIGAR_PTAP_FACTORED=0 TIGAR_PTAP_ELEMENTS=2"
run default timeout 2400 python tests/fuzz/fuzz_parity.py --seed $((S+1)) --cases 500
run poison env TIGAR_POOL_POISON=1 timeout 2400 python tests/fuzz/fuzz_parity.py --seed $((S+2)) --cases 300
run implicit env TIGAR_IMPLICIT_M=1 timeout 2400 python tests/fuzz/fuzz_parity.py --seed $((S+3)) --cases 400
run elements env $GEN timeout 2400 python tests/fuzz/fuzz_parity.py --seed $((S+4)) --cases 600
run elements_poison env $GEN TIGAR_POOL_POISON=1 timeout 2400 python tests/fuzz/fuzz_parity.py --seed $((S+5)) --cases 300
run element_chunks env $GEN TIGAR_IMPLICIT_M=1 TIGAR_ELEM_LAYERS=2 timeout 2400 python tests/fuzz/fuzz_parity.py --seed $((S+6)) --cases 500
run element_chunks_1 env $GEN TIGAR_IMPLICIT_M=1 TIGAR_ELEM_LAYERS=1 TIGAR_POOL_POISON=1 timeout 2400 python tests/fuzz/fuzz_parity.py --seed $((S+7)) --cases 300
run elements_alt env $GEN TIGAR_EL_MERGE=1 TIGAR_EL_VALU=1 TIGAR_EL_LISTS=1 timeout 2400 python tests/fuzz/fuzz_parity.py --seed $((S+8)) --cases 400
run symgrid env TIGAR_SPMV_SYM=2 TIGAR_KSP_PERSISTENT=0 timeout 2400 python tests/fuzz/fuzz_parity.py --seed $((S+9)) --cases 400 --max-rows 60000
run ranks timeout 2400 python tests/fuzz/fuzz_ranks.py --seed $((S+10)) --cases 40
run ranks_elements env $GEN TIGAR_ELEM_LAYERS=2 timeout 2400 python tests/fuzz/fuzz_ranks.py --seed $((S+11)) --cases 40
run sequences timeout 2400 python tests/fuzz/fuzz_sequences.py --seed $((S+12)) --cases 60
run sequences_elements env $GEN timeout 2400 python tests/fuzz/fuzz_sequences.py --seed $((S+13)) --cases 60
run kernels timeout 2400 python tests/fuzz/fuzz_kernels.py --seed $((S+14)) --cases 300
run newton timeout 2400 python tests/fuzz/fuzz_newton.py 60
run direct timeout 2400 python tests/fuzz/fuzz_direct.py --seed $((S+15)) --cases 150
run direct_poison env TIGAR_POOL_POISON=1 timeout 2400 python tests/fuzz/fuzz_direct.py --seed $((S+16)) --cases 100
run symgrid_kernels timeout 2400 python tests/fuzz/fuzz_symgrid.py --seed $((S+17)) --cases 60
cat $F
