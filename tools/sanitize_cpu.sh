#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (the oracle restatement and the front-ends' image I/O): sanitised
# builds of oracle/liboracle.so and optical-flow-1_amd/libofxio.so are swapped in for one run of the CPU tests that
# load them, then the normal builds are put back.  (GPU sanitizers are not available on the pool.)
set -e
R=/root/repo; T=$(mktemp -d)
SAN="-O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer"
gcc -std=c11 $SAN -fopenmp -ffp-contract=off -shared -o $T/liboracle.so $R/oracle/ofx_oracle.c -lm
gcc -std=gnu11 $SAN -I$R/include -I$R/optical-flow-1_amd/cli -shared -o $T/libofxio.so $R/optical-flow-1_amd/cli/ofx_io.c -ldl -lm
cp $R/oracle/liboracle.so $T/liboracle.orig; cp $R/optical-flow-1_amd/libofxio.so $T/libofxio.orig
restore() { cp $T/liboracle.orig $R/oracle/liboracle.so; cp $T/libofxio.orig $R/optical-flow-1_amd/libofxio.so; rm -rf $T; }
trap restore EXIT
cp $T/liboracle.so $R/oracle/liboracle.so; cp $T/libofxio.so $R/optical-flow-1_amd/libofxio.so
cd $R
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 OMP_NUM_THREADS=2 \
  timeout 1500 python -m pytest tests/test_oracle_golden.py tests/test_oracle_vs_ref.py tests/test_host_logic.py -x -q
