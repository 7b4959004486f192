#!/bin/bash
# rocprofv3 evidence for profiles/: kernel-trace stats of the bench command, then
# PMC passes (FETCH_SIZE and WRITE_SIZE need separate passes: TCC has 4 slots).
# usage: bash scripts/gpu_profile.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 50 --warmup 10 --no-cpu-baseline $@"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || { echo "trace run failed"; tail -5 $OUT/trace.err; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err || { echo "fetch run failed"; tail -5 $OUT/fetch.err; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err || { echo "write run failed"; tail -5 $OUT/write.err; }
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
# keep only small files for the merge back
find $OUT -name "*.csv" -size +3M -delete
ls -la $OUT $OUT/trace/* 2>/dev/null | head -30
