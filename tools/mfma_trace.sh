#!/bin/bash
# Where an element's time goes in the matrix-core stiffness kernel: builds it with -DFDD_MFMA_TRACE=1 (wave 0 of
# workgroup 0 accumulates the cycles between phase boundaries and prints the averages), runs the N = 15 microbenchmark
# once, then restores the normal build.  Through gpurun: bash tools/mfma_trace.sh
set -e
cd "$(dirname "$0")/.."
C=polynomial_reduction_with_full_domain_decomposition_preconditioner_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -DFDD_MFMA_TRACE=1 -c $C/fdd_stiffness_mfma.hip -o $C/build/fdd_stiffness_mfma.o
make -C $C -s
python tools/microbench.py --N 15 --E 32 --only stiffness 2>/dev/null | grep -E "mfma" | sort | uniq -c | sort -rn | head -8
touch $C/fdd_stiffness_mfma.hip
make -C $C -s
