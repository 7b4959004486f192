for lg in 17 18 19; do
  n=$((1<<lg))
  args=("X=1")
  for blk in 256 512; do for nc in 256 512 768 1024; do
    args+=("MI355_SPMV_BLOCK=$blk MI355_SPMV_ROWS_PER_CHUNK=$((n/nc))")
  done; done
  args+=("X=1")
  echo "== rows 2^$lg"
  timeout -k 10 400 bash scripts/gpu_knobs.sh "--rows-log2 $lg --kind vector" "${args[@]}"
done
