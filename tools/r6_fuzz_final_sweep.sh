# sweep of round 6 on its tree: fresh seeds, the switch sets of tests/test_gpu_fuzz.py and the new paths (element chunks)
O=gpurun_out/r6fuzz; mkdir -p $O; F=$O/r6_fuzz_sweep.txt; : > $F
run() { n=$1; shift; echo "== $n: $*" >> $F; ( "$@" 2>&1 | grep "^{" | tail -1 ) >> $F; }
GEN="TIGAR_PTAP_TENSOR=0 TIGAR_PTAP_FACTORED=0 TIGAR_PTAP_ELEMENTS=2"
run default timeout 2400 python tests/fuzz/fuzz_parity.py --seed 6901 --cases 500
run poison env TIGAR_POOL_POISON=1 timeout 2400 python tests/fuzz/fuzz_parity.py --seed 6902 --cases 300
run implicit env TIGAR_IMPLICIT_M=1 timeout 2400 python tests/fuzz/fuzz_parity.py --seed 6903 --cases 400
run elements env $GEN timeout 2400 python tests/fuzz/fuzz_parity.py --seed 6904 --cases 600
run elements_poison env $GEN TIGAR_POOL_POISON=1 timeout 2400 python tests/fuzz/fuzz_parity.py --seed 6905 --cases 300
run element_chunks env $GEN TIGAR_IMPLICIT_M=1 TIGAR_ELEM_LAYERS=2 timeout 2400 python tests/fuzz/fuzz_parity.py --seed 6906 --cases 500
run element_chunks_1 env $GEN TIGAR_IMPLICIT_M=1 TIGAR_ELEM_LAYERS=1 TIGAR_POOL_POISON=1 timeout 2400 python tests/fuzz/fuzz_parity.py --seed 6907 --cases 300
run elements_alt env $GEN TIGAR_EL_MERGE=1 TIGAR_EL_VALU=1 TIGAR_EL_LISTS=1 timeout 2400 python tests/fuzz/fuzz_parity.py --seed 6908 --cases 400
run symgrid env TIGAR_SPMV_SYM=2 TIGAR_KSP_PERSISTENT=0 timeout 2400 python tests/fuzz/fuzz_parity.py --seed 6909 --cases 400 --max-rows 60000
run ranks timeout 2400 python tests/fuzz/fuzz_ranks.py --seed 6910 --cases 40
run ranks_elements env $GEN TIGAR_ELEM_LAYERS=2 timeout 2400 python tests/fuzz/fuzz_ranks.py --seed 6911 --cases 40
run sequences timeout 2400 python tests/fuzz/fuzz_sequences.py --seed 6912 --cases 60
run sequences_elements env $GEN timeout 2400 python tests/fuzz/fuzz_sequences.py --seed 6913 --cases 60
run kernels timeout 2400 python tests/fuzz/fuzz_kernels.py --seed 6914 --cases 300
run newton timeout 2400 python tests/fuzz/fuzz_newton.py 60
cat $F
