#!/bin/bash
# Round-2 opening measurements (one gpurun call): gather ceiling, the new full-size config tests,
# PMC evidence for the gather-bound configs, a fresh trace of the merge kernels.
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_base
rm -rf $O; mkdir -p $O
echo "== gather ceiling"; timeout -k 10 300 $R/tools/bin/exp_gather > $O/exp_gather.txt 2>&1; tail -3 $O/exp_gather.txt
echo "== config tests"; cd $R && timeout -k 10 900 python -m pytest tests/test_gpu_configs.py -x -q -k "not eight" > $O/pytest_configs.log 2>&1; tail -5 $O/pytest_configs.log
cd /tmp
(rocprofv3 -L > $O/counters_avail.txt 2>&1 || true)
pmc() {  # tag counters bench-args...
  local tag=$1 ctr="$2"; shift; shift
  rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/pmc_$tag.json 2> $O/pmc_$tag.err || { echo "pmc $tag failed"; tail -3 $O/pmc_$tag.err; }
}
trace() {
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $O/trace_$tag.json 2> $O/trace_$tag.err || { echo "trace $tag failed"; tail -3 $O/trace_$tag.err; }
}
echo "== c5 merge"
trace c5_merge --workload c5-rmat24 --kind merge
pmc c5_merge_fetch FETCH_SIZE --workload c5-rmat24 --kind merge
pmc c5_merge_write WRITE_SIZE --workload c5-rmat24 --kind merge
pmc c5_merge_l2 "TCC_HIT_sum TCC_MISS_sum" --workload c5-rmat24 --kind merge
pmc c5_merge_req "TCC_REQ_sum TCC_EA0_RDREQ_sum" --workload c5-rmat24 --kind merge
echo "== s32-rand vector"
trace rand_vector --workload s32-rand --kind vector
pmc rand_vector_fetch FETCH_SIZE --workload s32-rand --kind vector
pmc rand_vector_l2 "TCC_HIT_sum TCC_MISS_sum" --workload s32-rand --kind vector
echo "== merge on the target (final code)"
trace s32_merge --kind merge
pmc s32_merge_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" --kind merge
pmc s32_merge_busy "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --kind merge
pmc s32_merge_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" --kind merge
cd $R
python3 scripts/summarize_pmc.py $O > $O/summary.txt 2>&1; cat $O/summary.txt
find $O -name "*.csv" -size +2M -delete
find $O -name "*.db" -delete
