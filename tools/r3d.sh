python tools/tt2_bench.py both 2>&1 | grep "per call"
for ex in 8 16 32 64; do for ey in 4 8 16; do TIGAR_TT2_ECH_X=$ex TIGAR_TT2_ECH_Y=$ey python tools/tt2_bench.py cfg4 2>&1 | grep "plan.ptap"; done; done
for ex in 6 12 24; do for ey in 6 12; do TIGAR_TT2_ECH_X=$ex TIGAR_TT2_ECH_Y=$ey python tools/tt2_bench.py cfg5 2>&1 | grep "plan.ptap"; done; done
python -c "
import cProfile, pstats, sys, os
sys.argv=['x','cfg4']
sys.path.insert(0,'tools')
import tt2_bench
cProfile.run('tt2_bench.run(4,256,1,reps=200)', '/tmp/prof.out')
pstats.Stats('/tmp/prof.out').sort_stats('cumulative').print_stats(18)
" 2>&1 | tail -40
