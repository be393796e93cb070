#!/bin/bash
# The host layer (composite setup, solvers, AMG build, mesh-file reader) under AddressSanitizer on the CPU build:
# tests/cpu_shim rebuilt with -fsanitize=address, the gloo tests run with libasan preloaded, then the normal build
# restored.  GPU sanitizers are not available on the pool; this covers the host-side C++ (most of the round-2 code).
set -e
cd "$(dirname "$0")/.."
S=tests/cpu_shim
H=polynomial_reduction_with_full_domain_decomposition_preconditioner_amd/host
A="-fsanitize=address -fno-omit-frame-pointer -g -O1"
make -C oracle -s
mkdir -p $S/_build
gcc $A -ffp-contract=off -fPIC -std=gnu99 -shared -Iinclude -Ioracle -o $S/_build/libfdd_cpu_shim.so $S/fdd_cpu_shim.c oracle/fdd_oracle_kernels.c -lm
g++ $A -std=c++17 -fPIC -shared -Iinclude -I$H -o $S/_build/libfdd_host_cpu.so $H/fdd_host_capi.cpp -L$S/_build -lfdd_cpu_shim -ldl -Wl,-rpath,'$ORIGIN'
# libstdc++ is preloaded with libasan: the interpreter is a C program, and the sanitizer's __cxa_throw interceptor must find
# the real one when it initialises (a rank thread that throws -- test_a_failing_rank_thread_... -- aborts in the check otherwise)
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_cpu_multirank.py tests/test_cpu_amg.py -x -q -p no:cacheprovider "$@" || rc=$?
rm -rf $S/_build && make -C $S -s
exit ${rc:-0}
