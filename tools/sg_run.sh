#!/bin/bash
# symgrid development loop on the GPU box: tests, then cfg3 bench + kernel stats
mkdir -p gpurun_out/sg; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_symgrid.py -q -x 2>&1 | tail -5
TIGAR_TRACE=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --companion 0 --live-traffic 0 --mapped-companion 0 > gpurun_out/sg/bench_cfg3.json 2> gpurun_out/sg/bench_cfg3.err
grep "symgrid\|cg:" gpurun_out/sg/bench_cfg3.err | tail -3
python -c "
import json
d=json.load(open('gpurun_out/sg/bench_cfg3.json'))
print(d['ms_per_step'], d['value'], d['config']['stages_s'], d['config']['cg_iterations'], d['roofline']['avg_launch_ms'])
"
rm -rf /tmp/sgprof; rocprofv3 --kernel-trace --stats -d /tmp/sgprof -o cfg3 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --companion 0 --live-traffic 0 --mapped-companion 0 > /dev/null 2>&1
python tools/rocpd_stats.py /tmp/sgprof/cfg3_results.db 12 | tee gpurun_out/sg/kstats.txt
