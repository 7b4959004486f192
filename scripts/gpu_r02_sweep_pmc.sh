#!/bin/bash
# SQ / LDS counters of the sweep kernel on a 65 537-column band (2^22 rows x 32, fp32), separate --pmc passes
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02_sweep_pmc
rm -rf $O; mkdir -p $O
cd /tmp
pmc() {
  local tag=$1 ctr="$2"; shift; shift
  rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/pmc_$tag.json 2> $O/pmc_$tag.err || { echo "pmc $tag failed"; tail -3 $O/pmc_$tag.err; }
}
pmc sweep_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" --band-half-width 32768 --kind vector
pmc sweep_busy "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --band-half-width 32768 --kind vector
pmc sweep_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" --band-half-width 32768 --kind vector
pmc sweep_l2 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" --band-half-width 32768 --kind vector
pmc window_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" --kind vector
pmc window_busy "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --kind vector
cd $R
python3 scripts/summarize_pmc.py $O > $O/summary.json 2> $O/summary.err
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat $O/summary.json
