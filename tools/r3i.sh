timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_api.py tests/test_gpu_configs.py -q -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -3
for it in 8 16 32 64; do echo iters $it; TIGAR_EXTRACT_ITERS=$it TIGAR_EXTRACT_KRON=0 python tools/kernel_trace.py --sum -- python bench.py --workload cfg2 --steps 5 --warmup 1 --no-cpu-baseline --companion 0 2>&1 | grep extract; done
echo auto; TIGAR_EXTRACT_KRON=0 python tools/kernel_trace.py --sum -- python bench.py --workload cfg2 --steps 5 --warmup 1 --no-cpu-baseline --companion 0 2>&1 | grep extract
