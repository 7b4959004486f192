#!/bin/bash
# C4 (fp64, 64-bit offsets, 27-point stencil stand-in): which plan knobs move it?
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_c4knobs.txt
: > $O
cd $R
run() { echo -n "$* : " >> $O; env "$@" timeout -k 10 120 python bench.py --workload c4-nlpkkt --kind ${KIND:-vector} --no-cpu-baseline --steps 40 --warmup 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('%8.1f us  win %s seg %s lanes %s grid %s' % (d['roofline']['kernel_ms']*1e3, c['x_window_elems'], c['x_window_segments'], c['lanes_per_row'], c['grid_blocks']))" >> $O 2>&1 || echo failed >> $O; }
run X=1
for r in 256 384 512 640 768 896 1024 1280; do run MI355_SPMV_ROWS_PER_CHUNK=$r; done
for t in 4 8 16 32; do run MI355_SPMV_LANES=$t; done
run MI355_SPMV_SEGMENTS=0
run MI355_SPMV_WINDOW=0
run MI355_SPMV_WINDOW=0 MI355_SPMV_LANES=4
run MI355_SPMV_WINDOW=0 MI355_SPMV_LANES=16
run MI355_SPMV_WINDOW=0 MI355_SPMV_LANES=32
KIND=light run X=1
KIND=merge run X=1
KIND=merge run MI355_MERGE_TPS=4
KIND=merge run MI355_MERGE_TPS=8
cat $O
