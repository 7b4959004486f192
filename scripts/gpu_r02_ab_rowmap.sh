#!/bin/bash
# A/B on one box: the new row mapping (an instruction reads consecutive rows) vs round 1's (a vector owns adjacent rows)
R=$GRAFT_REPO_ROOT; cd $R
A="MI355_SPMV_LIB=$R/spmv-samples_amd/lib/libmi355spmv.so"
B="MI355_SPMV_LIB=$R/spmv-samples_amd/lib_b/libmi355spmv.so"
{
echo "A = consecutive rows per instruction (new), B = adjacent rows per vector (round 1)"
echo "== s32-band fp32 (target)"; bash scripts/gpu_ab.sh "--kind vector" $A $B
echo "== s32-band fp64 i64"; bash scripts/gpu_ab.sh "--kind vector --s32-values f64 --s32-offsets i64" $A $B
echo "== c4"; bash scripts/gpu_ab.sh "--kind vector --workload c4-nlpkkt" $A $B
echo "== c2"; bash scripts/gpu_ab.sh "--kind vector --workload c2-cant" $A $B
echo "== c3"; bash scripts/gpu_ab.sh "--kind vector --workload c3-webgoogle" $A $B
echo "== light s32"; bash scripts/gpu_ab.sh "--kind light" $A $B
} 2>&1 | tee gpurun_out/r02_ab_rowmap.txt
