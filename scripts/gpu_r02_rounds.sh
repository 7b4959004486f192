#!/bin/bash
# few-round grids: does a shorter chunk (more, desynchronised rounds) beat the 2 048-row chunk at 2^19..2^21 rows?
for lg in 19 20 21 22; do for r in 0 512 768 1024 1536; do
  MI355_SPMV_ROWS_PER_CHUNK=$r python bench.py --no-cpu-baseline --steps 100 --warmup 20 --kind vector --rows-log2 $lg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d['config']; print('2^$lg rows/chunk knob $r: grid', c['grid_blocks'], 'window', c['x_window_elems'], round(r['kernel_ms']*1e3,1), 'us', round(r['achieved']), 'GB/s')"
done; done
