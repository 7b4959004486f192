# kernel-trace durations on the cant stand-in: the default plan against the plain one-pass kernel at 16 / 32 / 64 lanes per row
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cant_prof; rm -rf $O; mkdir -p $O
for env in "X=1" "MI355_SPMV_PLAIN=1 MI355_SPMV_LANES=16" "MI355_SPMV_PLAIN=1 MI355_SPMV_LANES=32" "MI355_SPMV_PLAIN=1 MI355_SPMV_LANES=64"; do
  tag=$(echo "$env" | tr ' =' '__')
  env $env rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 $R/bench.py --workload c2-cant --kind vector --steps 400 --warmup 200 --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
  f=$(find $O/$tag -name "*kernel_stats.csv" | head -1)
  echo "cant $env: $(grep csr_vector "$f" | cut -d, -f1-7 | cut -c1-60,200-400)"
  grep csr_vector "$f" | awk -F'","' '{print "   avg", $4, "min", $6}'
done
tag=small14
rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 $R/bench.py --rows-log2 14 --kind vector --steps 400 --warmup 200 --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
f=$(find $O/$tag -name "*kernel_stats.csv" | head -1); grep csr_vector "$f" | awk -F'","' '{print "2^14 default:", substr($1,2,50), "avg", $4, "min", $6}'
find $O -name "*.csv" -size +200k -delete
