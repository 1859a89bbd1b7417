#!/bin/bash
# rocprofv3 kernel statistics of one command (developer tool): tools/prof_run.sh <pattern> -- <command ...>
pat=$1; shift; shift
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt && rocprofv3 --kernel-trace --stats -d /tmp/kt -o k --output-format csv -- "$@" > /tmp/prof_run.log 2>&1
tail -3 /tmp/prof_run.log | cut -c1-1500
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/kstats.py $f "$pat"
