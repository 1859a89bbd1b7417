# targeted sweep after the fix of the box kernel's flag buffer (csrc/tg_ptap_box.hip): repeated knots in every direction and
# couplings added by hand, the detour of the first fix switched off (the default now)
S=${1:-7500}; O=gpurun_out/r6fuzz; mkdir -p $O; F=$O/r6_fuzz_box_sweep_$S.txt; : > $F
run() { n=$1; shift; echo "== $n: $*" >> $F; ( "$@" 2>&1 | grep "^{\|^FAIL" | tail -4 | cut -c1-700 ) >> $F; }
X='{"matrix": "random_extra"}'
run drops_extra timeout 2400 python tests/fuzz/fuzz_parity.py --seed $((S+1)) --cases 500 --drop-all --force "$X"
run drops_extra_implicit env TIGAR_IMPLICIT_M=1 timeout 2400 python tests/fuzz/fuzz_parity.py --seed $((S+2)) --cases 400 --drop-all --force "$X"
run drops_extra_poison env TIGAR_POOL_POISON=1 timeout 2400 python tests/fuzz/fuzz_parity.py --seed $((S+3)) --cases 300 --drop-all --force "$X"
run drops_extra_oneshot env TIGAR_PTAP_GROUPS="0,1,2" timeout 2400 python tests/fuzz/fuzz_parity.py --seed $((S+4)) --cases 300 --drop-all --force "$X"
run drops_extra_xy_z env TIGAR_PTAP_GROUPS="0,1;2" timeout 2400 python tests/fuzz/fuzz_parity.py --seed $((S+5)) --cases 300 --drop-all --force "$X"
run drops_any timeout 2400 python tests/fuzz/fuzz_parity.py --seed $((S+6)) --cases 400 --drop-all
run default timeout 2400 python tests/fuzz/fuzz_parity.py --seed $((S+7)) --cases 400
run sequences timeout 2400 python tests/fuzz/fuzz_sequences.py --seed $((S+8)) --cases 60
run ranks timeout 2400 python tests/fuzz/fuzz_ranks.py --seed $((S+9)) --cases 40
cat $F
