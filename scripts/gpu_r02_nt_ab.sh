#!/bin/bash
# A/B: nontemporal (library) vs plain (lib_plain, -DMI355_STREAM_PLAIN) loads of the Aj / Ax stream, power-law configs
for w in c5-rmat24 c3-webgoogle; do for k in vector light merge; do for lib in lib lib_plain lib lib_plain; do
  MI355_SPMV_LIB=$PWD/spmv-samples_amd/$lib/libmi355spmv.so python bench.py --no-cpu-baseline --steps 40 --warmup 5 --workload $w --kind $k 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$w $k $lib', round(r['kernel_ms']*1e3,2), 'us  min', round(r['kernel_ms_min']*1e3,2), 'median', round(r['kernel_ms_median']*1e3,2))"
done; done; done
