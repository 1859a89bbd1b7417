# round-5 random campaign on the new paths (fresh seeds); output under gpurun_out/r5fuzz
O=gpurun_out/r5fuzz; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*" >> $O/r5_fuzz_campaign.txt; ( "$@" 2>&1 | grep "^{" | tail -1 ) >> $O/r5_fuzz_campaign.txt; }
E="env TIGAR_PTAP_TENSOR=0 TIGAR_PTAP_FACTORED=0 TIGAR_PTAP_ELEMENTS=2"
run elements_1 $E timeout 1500 python tests/fuzz/fuzz_parity.py --seed 5101 --cases 400
run elements_2 $E TIGAR_POOL_POISON=1 timeout 1500 python tests/fuzz/fuzz_parity.py --seed 5102 --cases 300
run elements_seq $E timeout 1200 python tests/fuzz/fuzz_sequences.py --seed 5103 --cases 40
run symgrid_1 env TIGAR_SPMV_SYM=2 TIGAR_KSP_PERSISTENT=0 timeout 1500 python tests/fuzz/fuzz_parity.py --seed 5201 --cases 400
run symgrid_big env TIGAR_SPMV_SYM=2 TIGAR_KSP_PERSISTENT=0 timeout 1500 python tests/fuzz/fuzz_parity.py --seed 5202 --cases 60 --max-rows 400000
run symgrid_ranks env TIGAR_SPMV_SYM=2 TIGAR_KSP_PERSISTENT=0 timeout 1800 python tests/fuzz/fuzz_ranks.py --seed 5203 --cases 30
run default_1 timeout 1500 python tests/fuzz/fuzz_parity.py --seed 5301 --cases 400
run implicit_1 env TIGAR_IMPLICIT_M=1 timeout 1500 python tests/fuzz/fuzz_parity.py --seed 5302 --cases 300
run assembly timeout 1500 python tests/fuzz/fuzz_assembly.py --seed 5303 --cases 60
run newton timeout 1500 python tests/fuzz/fuzz_newton.py --seed 5304 --cases 30
run kernels timeout 1500 python tests/fuzz/fuzz_kernels.py --seed 5305 --cases 200
cat $O/r5_fuzz_campaign.txt
