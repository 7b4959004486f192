#!/bin/bash
# SQ counter pass on a bench command (kernel-trace/stats must NOT be combined with --pmc)
# usage: bash scripts/gpu_pmc.sh <tag> "<counters>" [bench args...]
set -o pipefail
TAG=$1; CTR="$2"; shift; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc $CTR --output-format csv -d $OUT/pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.log || { echo "pmc run failed"; tail -5 $OUT/err.log; }
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f[0])):
    n = row["Kernel_Name"].split("(")[0].split("<")[0].split("::")[-1]
    if n.startswith(("merge_", "csr_vector", "light_rows", "probe")):
        acc[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
find $OUT -name "*.csv" -size +3M -delete
