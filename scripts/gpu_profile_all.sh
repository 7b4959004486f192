#!/bin/bash
# kernel-trace evidence for every config/kind pair quoted in DESIGN.md §3.5 -> gpurun_out/trace_<tag>/
bash scripts/gpu_trace.sh s32_light --kind light
bash scripts/gpu_trace.sh s32_merge --kind merge
bash scripts/gpu_trace.sh c2_vector --workload c2-cant --kind vector
bash scripts/gpu_trace.sh c3_merge --workload c3-webgoogle --kind merge
bash scripts/gpu_trace.sh c4_vector --workload c4-nlpkkt --kind vector
bash scripts/gpu_trace.sh c4_merge --workload c4-nlpkkt --kind merge
bash scripts/gpu_trace.sh c5_merge --workload c5-rmat24 --kind merge
