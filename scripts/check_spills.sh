#!/bin/bash
# Fails if any kernel of libmi355spmv.so uses scratch memory (register spills): rebuilds every translation unit
# with -Rpass-analysis=kernel-resource-usage and reads the remarks.  Writes profiles/<tag>_kernel_resources.txt
# when a tag is given.   usage: bash scripts/check_spills.sh [tag]
set -o pipefail
cd "$(dirname "$0")/.." || exit 2
python3 scripts/check_spills.py "$@"
