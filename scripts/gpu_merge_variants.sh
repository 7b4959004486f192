#!/bin/bash
# merge-kind tuning sweep: IPT x window x workloads
mkdir -p gpurun_out
OUT=gpurun_out/merge_variants.txt
: > $OUT
for w in s32-band c2-cant c3-webgoogle c4-nlpkkt c5-rmat24; do
 for ipt in 8 16; do
  for win in 0 1; do
   r=$(MI355_MERGE_IPT=$ipt MI355_MERGE_WINDOW=$win timeout -k 10 200 python bench.py --workload $w --kind merge --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%9.1f us %7.0f GB/s' % (d['roofline']['kernel_ms']*1e3, d['achieved_hbm_gbps']))")
   echo "$w ipt=$ipt window=$win : $r" | tee -a $OUT
  done
 done
done
