mkdir -p gpurun_out/r3f
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r3f/cfg3.json 2> gpurun_out/r3f/cfg3.log
TIGAR_PTAP_FUSED=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --companion 0 > gpurun_out/r3f/cfg3_unfused.json 2> gpurun_out/r3f/cfg3_unfused.log
grep -h "stages\|self-check\|nodal" gpurun_out/r3f/*.log | cut -c1-260
python -c "
import json
j=json.load(open('gpurun_out/r3f/cfg3.json'))
print(j['value'], j['ms_per_step'], j['config'].get('value_pattern_verified'), j['config'].get('ms_per_step_pattern_verified'), j['roofline']['frac'])
"
