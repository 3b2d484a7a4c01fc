#!/bin/bash
# CPU only: the host library and the oracle rebuilt with AddressSanitizer + UBSan (into gpurun_out/asan/, nothing in-tree changes) and
# the CPU test-suite run against them.  (GPU AddressSanitizer is not available on this pool; device code is covered by the parity tests.)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
O=$R/gpurun_out/asan
mkdir -p $O
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
g++ -O1 -std=c++17 -fPIC -ffp-contract=off -fopenmp $SAN -I$R/include -shared -o $O/libselhost.so $R/cuda_selection_criteria_amd/csrc/host/selection_host.cpp -lz
gcc -O1 -std=gnu11 -fPIC -fopenmp -ffp-contract=off $SAN -shared -o $O/liboracle.so $R/oracle/selection_oracle.c $R/oracle/build_sketch_oracle.c -lz -lm
cd $R
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 \
  SELHOST_LIB=$O/libselhost.so ORACLE_LIB=$O/liboracle.so OMP_NUM_THREADS=4 \
  python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider "$@" 2>&1 | tee $O/pytest_asan.log | tail -15
grep -c "runtime error\|ERROR: AddressSanitizer" $O/pytest_asan.log || true
