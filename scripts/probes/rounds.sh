# chunk length x number of rounds of the chip (512 slots: two 512-thread workgroups per CU) on the S32-band shape
for n in 1572864 2097152 2621440 3145728; do
  echo "== rows $n"
  timeout -k 10 400 bash scripts/gpu_knobs.sh "--rows $n --kind vector" "X=1" "MI355_SPMV_ROWS_PER_CHUNK=2048" "MI355_SPMV_ROWS_PER_CHUNK=1536" "MI355_SPMV_ROWS_PER_CHUNK=1024" "X=1"
done
