# second part of the round-5 random campaign (harnesses with fixed seeds and positional case counts + fresh seeds of the first part)
O=gpurun_out/r5fuzz; mkdir -p $O; F=$O/r5_fuzz_campaign_part2.txt; : > $F
run() { n=$1; shift; echo "== $n: $*" >> $F; ( "$@" 2>&1 | grep "^{\|failed\|FAIL\|ok" | tail -2 ) >> $F; }
run assembly timeout 1500 python tests/fuzz/fuzz_assembly.py 200
run newton timeout 1500 python tests/fuzz/fuzz_newton.py 60
run elasticity timeout 1500 python tests/fuzz/fuzz_elasticity.py 40
E="env TIGAR_PTAP_TENSOR=0 TIGAR_PTAP_FACTORED=0 TIGAR_PTAP_ELEMENTS=2"
run elements_3 $E TIGAR_PTAP_WAVE=1 timeout 1500 python tests/fuzz/fuzz_parity.py --seed 5111 --cases 300
run elements_roundtrip $E TIGAR_FUZZ_ROUNDTRIP=1 timeout 1500 python tests/fuzz/fuzz_parity.py --seed 5112 --cases 200
run symgrid_2 env TIGAR_SPMV_SYM=2 TIGAR_KSP_PERSISTENT=0 TIGAR_POOL_POISON=1 timeout 1500 python tests/fuzz/fuzz_parity.py --seed 5211 --cases 300
run symgrid_ranks2 env TIGAR_SPMV_SYM=2 TIGAR_KSP_PERSISTENT=0 timeout 1800 python tests/fuzz/fuzz_ranks.py --seed 5213 --cases 40
run ranks_default timeout 1800 python tests/fuzz/fuzz_ranks.py --seed 5214 --cases 30
run sequences timeout 1200 python tests/fuzz/fuzz_sequences.py --seed 5215 --cases 60
cat $F
