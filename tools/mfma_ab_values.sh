#!/bin/bash
# A/B of a compile-time VALUE of the matrix-core stiffness kernel on ONE box: tools/mfma_ab_values.sh FLAG v1 v2 ...
set -e
flag=$1; shift
cd "$(dirname "$0")/.."
C=polynomial_reduction_with_full_domain_decomposition_preconditioner_amd/csrc
for v in "$@" "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -D$flag=$v -c $C/fdd_stiffness_mfma.hip -o $C/build/fdd_stiffness_mfma.o
    make -C $C -s
    echo -n "$flag=$v: "; python tools/microbench.py --N 15 --E 32 --only stiffness 2>/dev/null | grep mfma
done
# leave the default build behind
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -c $C/fdd_stiffness_mfma.hip -o $C/build/fdd_stiffness_mfma.o
make -C $C -s
