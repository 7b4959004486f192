#!/bin/bash
# kernel-trace of the vendor comparison script: per-kernel device durations, ours and rocSPARSE's,
# free of host launch overhead.  usage: bash scripts/gpu_vendor_trace.sh <workload>...
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/vendor_trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/scripts/gpu_vendor_cmp.py "$@" > $OUT/cmp.log 2> $OUT/trace.err || { echo "trace run failed"; tail -5 $OUT/trace.err; }
cd $GRAFT_REPO_ROOT
grep -v Warning $OUT/cmp.log | tail -5
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/vendor_trace/trace/**/*kernel_trace.csv", recursive=True)
d = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 20: continue
    v.sort()
    print("%-90s n %4d  median %9.2f us  min %9.2f us" % (k[:90], len(v), v[len(v) // 2], v[0]))
PY
find $OUT -name "*.csv" -size +3M -delete
