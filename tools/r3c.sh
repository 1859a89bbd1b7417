mkdir -p gpurun_out/r3c
timeout 300 python bench.py --workload cfg4 --steps 5 --warmup 2 > gpurun_out/r3c/cfg4_lu.json 2> gpurun_out/r3c/cfg4_lu.log
timeout 300 python bench.py --workload cfg4 --solver cg --rtol 1e-10 --steps 5 --warmup 2 > gpurun_out/r3c/cfg4_cg.json 2> gpurun_out/r3c/cfg4_cg.log
timeout 300 python bench.py --workload cfg5 --rtol 1e-10 --steps 5 --warmup 2 > gpurun_out/r3c/cfg5.json 2> gpurun_out/r3c/cfg5.log
grep -h "stages" gpurun_out/r3c/*.log | cut -c1-260
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof4 -o p -- python $GRAFT_REPO_ROOT/bench.py --workload cfg4 --solver cg --rtol 1e-10 --steps 3 --warmup 1 --companion 0 > /dev/null 2>&1
python - <<'PY'
import glob,csv
for f in glob.glob('/tmp/prof4/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print(r['Name'][:60], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
