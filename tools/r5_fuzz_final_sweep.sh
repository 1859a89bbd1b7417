# final sweep of the round on the final tree: fresh seeds, the switch sets of tests/test_gpu_fuzz.py and the new paths
O=gpurun_out/r5fuzz; mkdir -p $O; F=$O/r5_fuzz_final_sweep.txt; : > $F
run() { n=$1; shift; echo "== $n: $*" >> $F; ( "$@" 2>&1 | grep "^{" | tail -1 ) >> $F; }
run default timeout 2400 python tests/fuzz/fuzz_parity.py --seed 5901 --cases 700
run poison env TIGAR_POOL_POISON=1 timeout 2400 python tests/fuzz/fuzz_parity.py --seed 5902 --cases 400
run implicit env TIGAR_IMPLICIT_M=1 timeout 2400 python tests/fuzz/fuzz_parity.py --seed 5903 --cases 500
run no_tensor env TIGAR_PTAP_TENSOR=0 timeout 2400 python tests/fuzz/fuzz_parity.py --seed 5904 --cases 400
run general env TIGAR_PTAP_TENSOR=0 TIGAR_PTAP_FACTORED=0 timeout 2400 python tests/fuzz/fuzz_parity.py --seed 5905 --cases 400
run elements env TIGAR_PTAP_TENSOR=0 TIGAR_PTAP_FACTORED=0 TIGAR_PTAP_ELEMENTS=2 timeout 2400 python tests/fuzz/fuzz_parity.py --seed 5906 --cases 500
run symgrid env TIGAR_SPMV_SYM=2 TIGAR_KSP_PERSISTENT=0 timeout 2400 python tests/fuzz/fuzz_parity.py --seed 5907 --cases 500 --max-rows 60000
run persistent env TIGAR_KSP_PERSISTENT=1 timeout 2400 python tests/fuzz/fuzz_parity.py --seed 5908 --cases 300
run roundtrip env TIGAR_FUZZ_ROUNDTRIP=1 timeout 2400 python tests/fuzz/fuzz_parity.py --seed 5909 --cases 200
run ranks timeout 2400 python tests/fuzz/fuzz_ranks.py --seed 5910 --cases 50
run ranks_sym env TIGAR_SPMV_SYM=2 TIGAR_KSP_PERSISTENT=0 timeout 2400 python tests/fuzz/fuzz_ranks.py --seed 5911 --cases 50
run sequences timeout 2400 python tests/fuzz/fuzz_sequences.py --seed 5912 --cases 80
run kernels timeout 2400 python tests/fuzz/fuzz_kernels.py --seed 5913 --cases 400
run symgrid_tool timeout 2400 python tests/fuzz/fuzz_symgrid.py --seed 5914 --cases 300
run assembly timeout 2400 python tests/fuzz/fuzz_assembly.py 300
run newton timeout 2400 python tests/fuzz/fuzz_newton.py 80
cat $F
