#!/bin/bash
# Round 2, call G: bench.py launch modes (plain, self-spawned N=1 is plain, torchrun N=1, forced safety net), vendor column on THIS box
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
O=$R/gpurun_out/r02_g; rm -rf $O; mkdir -p $O
tr() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 1 "${@:2}"; }
echo "== torchrun N=1 native"; tr 29611 --steps 100 --warmup 10 --no-cpu-baseline > $O/tr1.json 2> $O/tr1.err; python -c "import json; d=json.loads(open('$O/tr1.json').read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['parallelism'])"
echo "== torchrun N=1 forced fallback, 4 sub-blocks"; MI355_BENCH_FORCE_FALLBACK=1 tr 29612 --steps 100 --warmup 10 --no-cpu-baseline --sub-blocks 4 > $O/tr1_fb.json 2> $O/tr1_fb.err; python -c "import json; d=json.loads(open('$O/tr1_fb.json').read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['parallelism'])" || tail -5 $O/tr1_fb.err
echo "== torchrun N=1 forced fallback c5"; MI355_BENCH_FORCE_FALLBACK=1 tr 29613 --steps 20 --warmup 3 --no-cpu-baseline --sub-blocks 2 --workload c5-rmat24 --kind light > $O/tr1_fb_c5.json 2> $O/tr1_fb_c5.err; python -c "import json; d=json.loads(open('$O/tr1_fb_c5.json').read()); print(d['ms_per_step'], d['scaling'], d['config']['parallelism'][:80])" || tail -5 $O/tr1_fb_c5.err
echo "== vendor column on this box"; ls tools/bin/libcmp_rocsparse.so && timeout -k 10 500 python scripts/gpu_vendor_cmp.py --no-torch-sparse s32-band s32-rand c2-cant c3-webgoogle c4-nlpkkt c5-rmat24 > $O/vendor.txt 2>&1; tail -30 $O/vendor.txt
