# developer experiments on the wave-per-row PtAP kernels (kernel times from rocprofv3)
run() { echo "== $*"; env "$@" python tools/kernel_trace.py --sum -- python tools/ptap_bench.py ${P:-3} ${N:-64} 2>&1 | grep -E "k_gw.*Li[12]E"; }
run TIGAR_X=0
run TIGAR_PTAP_WAVE_TILE1=193,193,8,8,8 TIGAR_PTAP_WAVE_TILE2=67,67,4,4,4
run TIGAR_PTAP_WAVE_TILE1=193,193,16,16,16 TIGAR_PTAP_WAVE_TILE2=67,67,8,8,8
run TIGAR_PTAP_WAVE_RPW1=32 TIGAR_PTAP_WAVE_RPW2=8
