#!/bin/bash
# Round 2, call E: light one-dequeue mode, vector round-robin on weight-cut plans, merge with two barriers, uniform chunk base.
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_e
rm -rf $O; mkdir -p $O
cd $R
echo "== gpu suite"; timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
echo "== sweep"; bash scripts/gpu_sweep.sh r02e > $O/sweep.txt 2>&1; cat $O/sweep.txt; cp gpurun_out/sweep_r02e.jsonl $O/
cd /tmp
pmc() {  # tag counters bench-args...
  local tag=$1 ctr="$2"; shift; shift
  rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/pmc_$tag.json 2> $O/pmc_$tag.err || { echo "pmc $tag failed"; tail -3 $O/pmc_$tag.err; }
}
echo "== c4 vector counters"
pmc c4_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" --workload c4-nlpkkt --kind vector
pmc c4_busy "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --workload c4-nlpkkt --kind vector
pmc c4_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" --workload c4-nlpkkt --kind vector
pmc c4_fetch FETCH_SIZE --workload c4-nlpkkt --kind vector
pmc c4_l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" --workload c4-nlpkkt --kind vector
pmc s32_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" --kind vector
pmc s32_busy "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --kind vector
pmc s32_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" --kind vector
cd $R
python3 scripts/summarize_pmc.py $O > $O/summary.txt 2>&1; cat $O/summary.txt
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
echo done
