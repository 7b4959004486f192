#!/bin/bash
# A/B builds: scripts/build_variant.sh NAME "-DFLAG ..." tu1 [tu2 ...]
# -> spmv-samples_amd/lib_NAME/libmi355spmv.so: the named translation units (e.g. csr_vector_f64) rebuilt with the extra
# flags, every other object taken from lib/.  Run a bench against it with MI355_SPMV_LIB=.../lib_NAME/libmi355spmv.so.
# (lib_*/ is git-ignored and listed in .gpurunignore only when stale: variants travel to the GPU box.)
set -e
cd "$(dirname "$0")/../spmv-samples_amd/csrc"
NAME=$1; FLAGS=$2; shift; shift
OUT=../lib_$NAME
mkdir -p $OUT
make -s -j8 >/dev/null
cp -p ../lib/*.o $OUT/
for tu in "$@"; do rm -f $OUT/$tu.o; done
make -s -j8 OUT=$OUT HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $FLAGS" >/dev/null
ls -la $OUT/libmi355spmv.so
