# kernel-trace durations (not the Python loop's step time) of small matrices: default plan / the plain CSR-vector kernel / lanes
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/small_prof2; rm -rf $O; mkdir -p $O
for lg in 12 14 15 16 17; do for env in "X=1" "MI355_SPMV_PLAIN=1" "MI355_SPMV_PLAIN=1 MI355_SPMV_LANES=16" "MI355_SPMV_PLAIN=1 MI355_SPMV_LANES=32"; do
  tag=$(echo "${lg}_$env" | tr ' =' '__')
  env $env rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 $R/bench.py --rows-log2 $lg --kind vector --steps 300 --warmup 100 --no-cpu-baseline > $O/$tag.json 2> $O/$tag.err
  f=$(find $O/$tag -name "*kernel_stats.csv" | head -1)
  echo "2^$lg $env: $(grep -E 'csr_vector' "$f" | awk -F'","' '{printf "%s calls %s avg %s min %s\n", substr($1,2,60), $2, $4, $6}' | head -2 | tr '\n' ' ')"
done; done
find $O -name "*.csv" -size +200k -delete
