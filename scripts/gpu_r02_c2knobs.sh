#!/bin/bash
# C2 (cant stand-in, 33 MB, cache-resident, latency-bound): which plan knobs move it?  kernel time from the rocprofv3 trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd /tmp
O=$R/gpurun_out/r02_c2knobs.txt; : > $O
run() { local tag="$*"; rm -rf /tmp/c2k; env "$@" rocprofv3 --kernel-trace --output-format csv -d /tmp/c2k -- python3 $R/bench.py --workload c2-cant --kind ${KIND:-vector} --no-cpu-baseline --steps 300 --warmup 50 > /tmp/c2k.json 2>/dev/null
  python3 - "$tag" >> $O <<'P'
import csv,glob,sys,json,statistics
f=glob.glob('/tmp/c2k/**/*kernel_trace.csv',recursive=True)
d=[]
for r in csv.DictReader(open(f[0])):
    n=r['Kernel_Name']
    if 'window_kernel' in n or 'merge_tile' in n: d.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
d=d[len(d)//3:]
c=json.loads(open('/tmp/c2k.json').read())['config']
print("%-50s mean %6.2f median %6.2f min %6.2f  T=%s grid=%s win=%s"%(sys.argv[1],sum(d)/len(d),statistics.median(d),min(d),c['lanes_per_row'],c['grid_blocks'],c['x_window_elems']))
P
}
run X=1
run MI355_SPMV_LANES=32
run MI355_SPMV_LANES=8
run MI355_SPMV_LANES=32 MI355_SPMV_ROWS_PER_CHUNK=32
run MI355_SPMV_LANES=32 MI355_SPMV_ROWS_PER_CHUNK=64
run MI355_SPMV_LANES=32 MI355_SPMV_ROWS_PER_CHUNK=128
run MI355_SPMV_ROWS_PER_CHUNK=32
run MI355_SPMV_ROWS_PER_CHUNK=128
run MI355_SPMV_WINDOW=0
run MI355_SPMV_WINDOW=0 MI355_SPMV_LANES=32
run MI355_SPMV_WINDOW_FROM_BAND=0
KIND=light run X=1
KIND=light run MI355_SPMV_LANES=32
cat $O
