#!/bin/bash
# kernel-trace only (no PMC) of a bench command; prints per-kernel mean duration
# usage: bash scripts/gpu_trace.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-t}; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/trace.err || { echo "trace run failed"; tail -5 $OUT/trace.err; }
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_prof.py $OUT | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print('%-28s launches %4d  mean %9.2f us  min %9.2f us' % (k, v['launches'], v['trimmed_mean_us'], v['min_us']))
"
find $OUT -name "*.csv" -size +3M -delete
