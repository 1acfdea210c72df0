#!/bin/bash
# The CPU oracle under AddressSanitizer + UBSan (gcc): `make -C oracle asan`, then the oracle's own CPU tests against that build
# (HE_ORACLE_LIB_PATH).  Covers the per-thread scratch cache of he_oracle.c (scr_alloc / scr_free pairs) and the batch loops.
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
make -C $ROOT/oracle -s asan
cd $ROOT
HE_ORACLE_LIB_PATH=$ROOT/oracle/_build/libhe_oracle_asan.so LD_PRELOAD=$(gcc -print-file-name=libasan.so) \
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_oracle_kat.py tests/test_exact_model.py tests/test_sharding_gloo.py -x -q -m "not gpu" "$@"
