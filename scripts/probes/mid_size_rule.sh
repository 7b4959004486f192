# the mid-size single-round rule (analyze.hip, shape_chunks) against the 256-thread plan it replaces, per size / kind / type
for spec in "--rows-log2 17" "--rows-log2 18" "--rows-log2 19" "--rows-log2 20" "--rows-log2 18 --s32-values f64" "--rows-log2 19 --s32-values f64" "--rows-log2 18 --s32-offsets i64" "--workload c2-cant"; do
  for kind in vector light; do
    echo "== $spec $kind"
    timeout -k 10 300 bash scripts/gpu_knobs.sh "$spec --kind $kind" "X=1" "MI355_SPMV_BLOCK=256" "X=1" "MI355_SPMV_BLOCK=256"
  done
done
