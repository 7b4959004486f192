#!/bin/bash
# C2 (cant stand-in): workgroup size and chunk length against the default single round of 256-thread workgroups
for e in X=1 MI355_SPMV_BLOCK=512 "MI355_SPMV_BLOCK=512 MI355_SPMV_ROWS_PER_CHUNK=128" "MI355_SPMV_BLOCK=512 MI355_SPMV_ROWS_PER_CHUNK=256" "MI355_SPMV_ROWS_PER_CHUNK=96" "MI355_SPMV_ROWS_PER_CHUNK=128" "MI355_SPMV_ROWS_PER_CHUNK=256"; do
  env $e python bench.py --no-cpu-baseline --steps 3000 --warmup 1000 --kind vector --workload c2-cant 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print('$e :', 'T', c['lanes_per_row'], 'grid', c['grid_blocks'], 'window', c['x_window_elems'], round(d['roofline']['kernel_ms']*1e3,2), 'us')"
done
