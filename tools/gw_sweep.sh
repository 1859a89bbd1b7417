# developer experiments on the wave-per-row PtAP kernels (kernel times from rocprofv3)
run() { echo "== $*"; env "$@" python tools/kernel_trace.py --sum -- python tools/ptap_bench.py ${P:-3} ${N:-64} 2>&1 | grep -E "k_gw.*Li[12]E|ptap again"; }
run TIGAR_PTAP_WAVE_LG2=5
run TIGAR_PTAP_WAVE_LG2=4
run TIGAR_PTAP_WAVE_LG1=5
run TIGAR_PTAP_WAVE=0
