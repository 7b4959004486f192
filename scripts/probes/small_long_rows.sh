# where the plain one-pass kernel stops winning on LONGER rows (64 / 128 per row): kernel-trace durations, default plan against the plain kernel
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/small_long; rm -rf $O; mkdir -p $O
for spec in "--nnz-per-row 64 --rows-log2 16" "--nnz-per-row 64 --rows-log2 17" "--nnz-per-row 128 --rows-log2 15" "--nnz-per-row 128 --rows-log2 16" "--nnz-per-row 16 --rows-log2 18" "--nnz-per-row 8 --rows-log2 19"; do for env in "MI355_SPMV_SMALL=0" "MI355_SPMV_PLAIN=1 MI355_SPMV_LANES=16" "MI355_SPMV_PLAIN=1 MI355_SPMV_LANES=8" "MI355_SPMV_PLAIN=1 MI355_SPMV_LANES=32"; do
  tag=$(echo "${spec}_$env" | tr ' =-' '___')
  env $env rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 $R/bench.py $spec --kind vector --steps 300 --warmup 100 --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
  f=$(find $O/$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$spec | $env" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "csr_vector" in r["Name"]:
        print("%-72s %-40s avg %8.1f min %s" % (sys.argv[2], r["Name"].split("(")[0].replace("void mi355::","")[:40], float(r["AverageNs"]), r["MinNs"]))
PY
done; done
find $O -name "*.csv" -size +200k -delete
