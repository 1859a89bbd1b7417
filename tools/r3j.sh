timeout 900 python -m pytest tests/test_gpu_direct_solver.py tests/test_gpu_configs.py -q -x 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -4
timeout 300 python bench.py --workload cfg4 --steps 3 --warmup 1 2>&1 | grep stages
TIGAR_LU_FUSED=0 timeout 300 python bench.py --workload cfg4 --steps 3 --warmup 1 2>&1 | grep stages
