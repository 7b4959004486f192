#!/bin/bash
# Round 2, call H: deeper load pipeline for the R = 2 bodies (fp64; fp32 rows of 33+ nonzeros)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
echo "== gpu suite"; timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
echo "== default bench line count"; python bench.py --steps 50 --no-cpu-baseline 2>/dev/null | wc -l
echo "== sweep"; bash scripts/gpu_sweep.sh r02h 2>&1 | tail -7
cd /tmp
for w in c2-cant c4-nlpkkt; do for k in vector light; do
rm -rf /tmp/tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $R/bench.py --workload $w --kind $k --no-cpu-baseline --steps 300 --warmup 50 > /dev/null 2>&1
python3 - $w $k <<'P'
import csv,glob,sys,statistics
f=glob.glob('/tmp/tr/**/*kernel_trace.csv',recursive=True)
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f[0])) if 'window_kernel' in r['Kernel_Name']]
d=d[len(d)//3:]
print("trace %s %s: mean %.2f median %.2f min %.2f us"%(sys.argv[1],sys.argv[2],sum(d)/len(d),statistics.median(d),min(d)))
P
done; done
