#!/bin/bash
# Host-side code (bridge, client, parameter tables) under AddressSanitizer + UBSan on the CPU: builds a sanitised variant of the
# library from the same sources (device objects as built by the Makefile) and runs the CPU test-suite against it, selected with
# HE355_LIB_PATH: the product library is never touched.  GPU ASan is not available on the pool; the device code is covered by the
# parity tests instead.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC=$ROOT/reference-seal-backend_amd/csrc
CLANG=/opt/rocm/lib/llvm/bin/clang++
OUT=${TMPDIR:-/tmp}/he355_asan
mkdir -p $OUT
make -C $SRC -j6 > /dev/null
cd $SRC
for f in he_params.cpp bridge/*.cpp client/*.cpp; do
  $CLANG -O1 -g -std=c++17 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -c $f -o $OUT/$(echo $f | tr '/' '_').o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -o $OUT/libasan_backend.so $OUT/*.o _obj/he355_kernels.shoup.o _obj/he355_kernels.fold.o _obj/he355_kernels_client.shoup.o _obj/he355_kernels_client.fold.o _obj/he355_kernels_lds.shoup.o _obj/he355_kernels_lds.fold.o _obj/he355_api.o
cd $ROOT
HE355_LIB_PATH=$OUT/libasan_backend.so LD_PRELOAD=$($CLANG -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests -x -q -m "not gpu"
