#!/bin/bash
# The test-only lane simulator (tests/csim: the product's lane programs, client code and BEHZ per-coefficient arithmetic compiled for the
# CPU) under AddressSanitizer + UBSan: builds a sanitised libcsim.so, runs tests/test_lane_sim.py and tests/test_behz_sim_cpu.py against
# it, and restores the plain build.  CPU only.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/tests/csim"
make -s
cp _build/libcsim.so /tmp/libcsim_plain_$$.so
cp _build/libcsim_fold.so /tmp/libcsim_fold_plain_$$.so
trap 'cp /tmp/libcsim_plain_$$.so "$ROOT/tests/csim/_build/libcsim.so"; cp /tmp/libcsim_fold_plain_$$.so "$ROOT/tests/csim/_build/libcsim_fold.so"; rm -f /tmp/libcsim_plain_$$.so /tmp/libcsim_fold_plain_$$.so' EXIT
# both forms of the u64 engine (csrc/modarith.h): libcsim.so = Shoup quotients, libcsim_fold.so = fold reduction
for form in 0 1; do
  out=_build/libcsim.so; [ $form = 1 ] && out=_build/libcsim_fold.so
  g++ -O1 -g -std=c++17 -fPIC -mfma -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -Wno-unknown-pragmas -DHE355_U64_FOLD=$form -shared \
    -o $out sim_ntt.cpp sim_client.cpp sim_behz.cpp ../../reference-seal-backend_amd/csrc/he_params.cpp ../../reference-seal-backend_amd/csrc/client/he_client.cpp
done
touch _build/libcsim.so _build/libcsim_fold.so # (newer than the sources: the tests' own `make` must not rebuild them plain)
cd "$ROOT"
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_behz_sim_cpu.py tests/test_lane_sim.py -x -q
