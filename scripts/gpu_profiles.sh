#!/bin/bash
# The evidence for profiles/: kernel traces (rocprofv3 --kernel-trace --stats) and PMC passes (separate runs, as the
# microarch guide prescribes) of bench.py on every config / kind quoted in DESIGN.md, the vendor library on the same box,
# and the default bench line.   usage: bash scripts/gpu_profiles.sh [tag]      (one box, ~4 minutes)
set -o pipefail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
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
MI355_MERGE_ROWS=0 trace s32_merge_walk --kind merge --steps 60
trace c2_vector --workload c2-cant --kind vector --steps 400 --warmup 200
trace c2_light --workload c2-cant --kind light --steps 400 --warmup 200
trace c2_merge --workload c2-cant --kind merge --steps 400 --warmup 200
trace c3_merge --workload c3-webgoogle --kind merge --steps 300 --warmup 100
trace c3_vector --workload c3-webgoogle --kind vector
trace c4_vector --workload c4-nlpkkt --kind vector
trace c4_light --workload c4-nlpkkt --kind light
trace c4_merge --workload c4-nlpkkt --kind merge
trace c5_merge --workload c5-rmat24 --kind merge --steps 30 --warmup 20
trace c5_vector --workload c5-rmat24 --kind vector --steps 30 --warmup 20
trace c5_light --workload c5-rmat24 --kind light --steps 30 --warmup 20
trace rand_vector --workload s32-rand --kind vector --steps 40 --warmup 20
trace wide32k_vector --band-half-width 32768 --kind vector
trace wide32k_light --band-half-width 32768 --kind light
trace wide16k_vector --band-half-width 16384 --kind vector
echo "traces done"
for k in vector light merge; do
  pmc s32_${k}_fetch FETCH_SIZE --kind $k
  pmc s32_${k}_write WRITE_SIZE --kind $k
done
pmc wide32k_vector_fetch FETCH_SIZE --band-half-width 32768 --kind vector
pmc wide32k_vector_write WRITE_SIZE --band-half-width 32768 --kind vector
pmc c4_vector_fetch FETCH_SIZE --workload c4-nlpkkt --kind vector
pmc c4_vector_write WRITE_SIZE --workload c4-nlpkkt --kind vector
pmc c4_vector_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" --workload c4-nlpkkt --kind vector
pmc c4_vector_busy "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" --workload c4-nlpkkt --kind vector
pmc c4_vector_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" --workload c4-nlpkkt --kind vector
pmc s32_vector_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" --kind vector
pmc s32_vector_busy "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" --kind vector
pmc s32_vector_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" --kind vector
MI355_MERGE_ROWS=0 pmc s32_mergewalk_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" --kind merge
MI355_MERGE_ROWS=0 pmc s32_mergewalk_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" --kind merge
echo "pmc done"
cd $R
python3 scripts/summarize_pmc.py $O > $O/summary.json 2> $O/summary.err
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
python3 bench.py --cold --no-cpu-baseline --workload c2-cant --kind vector --steps 200 > $O/bench_c2_cold.json 2>/dev/null
python3 bench.py --no-cpu-baseline --workload c2-cant --kind vector --steps 2000 --warmup 200 > $O/bench_c2_warm.json 2>/dev/null
python3 scripts/probes/vendor_cmp.py s32-band c2-cant c3-webgoogle c4-nlpkkt c5-rmat24 s32-rand --no-torch-sparse > $O/vendor_rocsparse_same_box.txt 2> $O/vendor.err || tail -3 $O/vendor.err
# keep the small per-run stats tables, drop the bulky traces
for d in $O/trace_*/; do f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/$(basename $d)_kernel_stats.csv; done
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
head -c 6000 $O/summary.json
