#!/bin/bash
# CPU-side sanitizer pass (GPU ASan is not available on the pool): the oracle library and the product's Matrix Market loader
# built with -fsanitize=address,undefined, their test files run with the ASan runtime preloaded into Python.
# (The loader's exception tests are left out: a preloaded ASan cannot intercept __cxa_throw from a library loaded later —
#  "CHECK failed: real___cxa_throw != 0" is the tool's, not a finding.)  Restores the normal builds afterwards.
set -e
cd "$(dirname "$0")/.."   # (repo root)
FLAGS="-std=c++17 -O1 -g -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -Wall -Wextra"
cp oracle/liboracle.so /tmp/liboracle.so.bak; cp tests/cpp/libhostload.so /tmp/libhostload.so.bak 2>/dev/null || true
trap 'cp /tmp/liboracle.so.bak oracle/liboracle.so; [ -f /tmp/libhostload.so.bak ] && cp /tmp/libhostload.so.bak tests/cpp/libhostload.so' EXIT
g++ $FLAGS -o oracle/liboracle.so oracle/spmv_oracle.cpp -lpthread
(cd tests/cpp && g++ $FLAGS -o libhostload.so host_load_capi.cpp -lpthread)
ASAN=$(g++ -print-file-name=libasan.so)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_oracle.py -q -p no:cacheprovider
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_host_loader.py -q -p no:cacheprovider -k "not exceptions"
