#!/bin/bash
# A/B of a compile-time switch of the matrix-core stiffness kernel on ONE box (box-to-box variation is a few per cent):
# builds the object with -D<flag>=0 and =1 in turn and runs the N = 15 microbenchmark three times each.
# usage (through gpurun): bash tools/mfma_ab.sh FDD_MFMA_UNCOND_PREFETCH
set -e
flag=$1
cd "$(dirname "$0")/.."
C=polynomial_reduction_with_full_domain_decomposition_preconditioner_amd/csrc
for v in 0 1 0 1; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -D$flag=$v -c $C/fdd_stiffness_mfma.hip -o $C/build/fdd_stiffness_mfma.o
    make -C $C -s
    for i in 1 2; do echo -n "$flag=$v: "; python tools/microbench.py --N 15 --E 32 --only stiffness 2>/dev/null | grep mfma; done
done
