#!/bin/bash
# parity suite under forced code paths (window on/off, sampled vs band-placed window, no segments, merge run
# lengths and search widths, weight-cut chunks on/off, workgroup size, the 4-byte fallback kernel, long-row pass,
# the sweeping window forced / forbidden, merge's row-parallel runs and in-kernel search forced / forbidden, the
# 64-bit chunk path, forced vector widths)
mkdir -p gpurun_out
rc=0
skip=${1:-0}; i=0
for env in "MI355_SPMV_WINDOW=1" "MI355_SPMV_WINDOW=0" "MI355_SPMV_WINDOW_FROM_BAND=0" "MI355_SPMV_SEGMENTS=0" "MI355_MERGE_TPS=1" "MI355_MERGE_TPS=5" "MI355_MERGE_SEARCH_LANES=1" "MI355_MERGE_SEARCH_LANES=4" "MI355_SPMV_BALANCE=1" "MI355_SPMV_BALANCE=0" "MI355_SPMV_BLOCK=512" "MI355_SPMV_BLOCK=256" "MI355_SPMV_PLAIN=1" "MI355_SPMV_LONG_STEPS=1" "MI355_MERGE_BLOCK=512" "MI355_SPMV_GIANT=0" "MI355_SPMV_SWEEP=1" "MI355_SPMV_SWEEP=0" "MI355_MERGE_ROWS=1" "MI355_MERGE_ROWS=0" "MI355_MERGE_FUSED=0" "MI355_MERGE_FUSED=1" "MI355_SPMV_REL32_LIMIT=5000" "MI355_SPMV_LANES=4" "MI355_SPMV_LANES=32" "MI355_MERGE_WIDE_WINDOW=0" "MI355_MERGE_SEGMENTS=0" "MI355_SPMV_SMALL=1" "MI355_SPMV_GIANT_ROW=4096" "MI355_SPMV_PLAN_CACHE=0"; do
  i=$((i+1)); [ $i -le $skip ] && continue
  env $env timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/pt_env.log 2>&1
  r=$?
  echo "$env : exit $r : $(tail -1 gpurun_out/pt_env.log)"
  if [ $r -ne 0 ]; then rc=1; tail -30 gpurun_out/pt_env.log; break; fi
done
exit $rc
