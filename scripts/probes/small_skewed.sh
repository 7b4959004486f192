cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/small_skewed; rm -rf $O; mkdir -p $O
for sc in 12 14 16; do for env in "X=1" "MI355_SPMV_PLAIN=1"; do
  tag=$(echo "${sc}_$env" | tr ' =-' '___')
  env $env rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 $R/scripts/probes/small_skewed.py $sc > $O/$tag.log 2> $O/$tag.err
  f=$(find $O/$tag -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "rmat-$sc $env" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "mi355::" in r["Name"] and int(r["Calls"]) >= 100:
        print("%-28s %-52s calls %5s avg %8.1f min %s" % (sys.argv[2], r["Name"].split("(")[0].replace("void mi355::","")[:52], r["Calls"], float(r["AverageNs"]), r["MinNs"]))
PY
done; done
find $O -name "*.csv" -size +200k -delete
