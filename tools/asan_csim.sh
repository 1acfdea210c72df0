#!/bin/bash
# The test-only lane simulator (tests/csim: the product's lane programs, client code and BEHZ per-coefficient arithmetic compiled for the
# CPU) under AddressSanitizer + UBSan: builds a sanitised libcsim.so, runs tests/test_lane_sim.py and tests/test_behz_sim_cpu.py against
# it, and restores the plain build.  CPU only.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/tests/csim"
make -s
cp _build/libcsim.so /tmp/libcsim_plain_$$.so
trap 'cp /tmp/libcsim_plain_$$.so "$ROOT/tests/csim/_build/libcsim.so"; rm -f /tmp/libcsim_plain_$$.so' EXIT
g++ -O1 -g -std=c++17 -fPIC -mfma -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-unknown-pragmas -shared \
  -o _build/libcsim.so sim_ntt.cpp sim_client.cpp sim_behz.cpp ../../reference-seal-backend_amd/csrc/he_params.cpp ../../reference-seal-backend_amd/csrc/client/he_client.cpp
cd "$ROOT"
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_behz_sim_cpu.py tests/test_lane_sim.py -x -q
