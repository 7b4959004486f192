# the mid-size rule at other row lengths: ~8 M nonzeros (70 MB in fp32) and ~16 M at 8 / 16 / 64 / 128 per row
for spec in "--nnz-per-row 8 --rows-log2 20" "--nnz-per-row 16 --rows-log2 19" "--nnz-per-row 64 --rows-log2 17" "--nnz-per-row 128 --rows-log2 16" "--nnz-per-row 16 --rows-log2 20" "--nnz-per-row 64 --rows-log2 18" "--nnz-per-row 128 --rows-log2 17"; do
  echo "== $spec"
  timeout -k 10 300 bash scripts/gpu_knobs.sh "$spec --kind vector" "X=1" "MI355_SPMV_BLOCK=256" "X=1" "MI355_SPMV_BLOCK=256"
done
