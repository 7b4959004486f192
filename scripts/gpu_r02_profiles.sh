#!/bin/bash
# Round-2 evidence for profiles/: kernel traces (rocprofv3 --kernel-trace --stats) and PMC passes (separate runs)
# of bench.py on every config / kind quoted in DESIGN.md.   usage: bash scripts/gpu_r02_profiles.sh [tag]
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
O=$R/gpurun_out/${TAG}_prof
rm -rf $O; mkdir -p $O
cd /tmp
trace() {
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$tag -- python3 $R/bench.py --steps 100 --warmup 50 --no-cpu-baseline "$@" > $O/trace_$tag.json 2> $O/trace_$tag.err || { echo "trace $tag failed"; tail -3 $O/trace_$tag.err; }
}
pmc() {
  local tag=$1 ctr="$2"; shift; shift
  rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$tag -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $O/pmc_$tag.json 2> $O/pmc_$tag.err || { echo "pmc $tag failed"; tail -3 $O/pmc_$tag.err; }
}
trace s32_vector --kind vector
trace s32_light --kind light
trace s32_merge --kind merge
trace c2_vector --workload c2-cant --kind vector
trace c2_light --workload c2-cant --kind light
trace c2_merge --workload c2-cant --kind merge
trace c3_merge --workload c3-webgoogle --kind merge
trace c3_vector --workload c3-webgoogle --kind vector
trace c4_vector --workload c4-nlpkkt --kind vector
trace c4_merge --workload c4-nlpkkt --kind merge
trace c5_merge --workload c5-rmat24 --kind merge
trace c5_vector --workload c5-rmat24 --kind vector
trace c5_light --workload c5-rmat24 --kind light
trace rand_vector --workload s32-rand --kind vector
trace wide32k_vector --band-half-width 32768 --kind vector
trace wide16k_vector --band-half-width 16384 --kind vector
for k in vector light merge; do
  pmc s32_${k}_fetch FETCH_SIZE --kind $k
  pmc s32_${k}_write WRITE_SIZE --kind $k
done
pmc wide32k_vector_fetch FETCH_SIZE --band-half-width 32768 --kind vector
pmc wide32k_vector_write WRITE_SIZE --band-half-width 32768 --kind vector
pmc c4_vector_fetch FETCH_SIZE --workload c4-nlpkkt --kind vector
pmc c4_vector_write WRITE_SIZE --workload c4-nlpkkt --kind vector
pmc s32_merge_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" --kind merge
pmc s32_merge_busy "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --kind merge
pmc s32_merge_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" --kind merge
cd $R
python3 scripts/summarize_pmc.py $O > $O/summary.json 2> $O/summary.err
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --cold --no-cpu-baseline --workload c2-cant --kind vector --steps 200 > $O/bench_c2_cold.json 2>/dev/null
python3 bench.py --no-cpu-baseline --workload c2-cant --kind vector --steps 2000 --warmup 100 > $O/bench_c2_warm.json 2>/dev/null
# keep the small per-run stats tables, drop the bulky traces
for d in $O/trace_*/; do f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/$(basename $d)_kernel_stats.csv; done
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat $O/summary.json | head -150
