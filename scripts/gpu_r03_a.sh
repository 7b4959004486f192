#!/bin/bash
# Round 3, run A: (1) SQ / LDS counters of csr_vector_window_kernel on the C4 stand-in (fp64, i64) beside the same
# kernel on 32-per-row fp64 rows, (2) workgroup-size / chunk knobs on the C2 stand-in, (3) the baseline traces of this box.
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03_a
rm -rf $O; mkdir -p $O
cd /tmp
pmc() {
  local tag=$1 ctr="$2"; shift; shift
  rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/pmc_$tag.json 2> $O/pmc_$tag.err || { echo "pmc $tag failed"; tail -3 $O/pmc_$tag.err; }
}
trace() {
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -- python3 $R/bench.py --steps 100 --warmup 50 --no-cpu-baseline "$@" > $O/trace_$tag.json 2> $O/trace_$tag.err || { echo "trace $tag failed"; tail -3 $O/trace_$tag.err; }
}
for w in "c4 --workload c4-nlpkkt" "s32f64 --s32-values f64 --s32-offsets i64" "s32"; do
  set -- $w; tag=$1; shift
  pmc ${tag}_vector_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" --kind vector "$@"
  pmc ${tag}_vector_busy "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" --kind vector "$@"
  pmc ${tag}_vector_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU" --kind vector "$@"
  echo "pmc $tag done"
done
trace c4_vector --workload c4-nlpkkt --kind vector
trace s32_vector --kind vector
trace c2_vector --workload c2-cant --kind vector
trace c3_merge --workload c3-webgoogle --kind merge
echo traces done
cd $R
python3 scripts/summarize_pmc.py $O > $O/summary.json 2> $O/summary.err
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
bash scripts/gpu_r02_c2_block.sh > $O/c2_knobs.txt 2>&1
cat $O/c2_knobs.txt
cat $O/summary.json
