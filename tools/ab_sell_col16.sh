#!/bin/bash
# development A/B: the sliced-ELL copy's compact column form (FDD_TUNE_CSR_SELL_COL16: 0 = 32-bit columns in every slice) on the
# reference-default step, box and Kershaw
cd "$(dirname "$0")/.."
for mesh in box kershaw; do
  for c16 in 0 1 0 1; do
    FDD_TUNE_CSR_SELL_COL16=$c16 python3 bench.py --mesh $mesh --steps 6 --warmup 2 --no-time-to-tolerance --no-cpu-baseline --no-kershaw --no-stencil > gpurun_out/ab10.json 2>/dev/null
    python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/ab10.json') if l.startswith('{')][-1])
print('$mesh sell_col16 $c16: headline %.3f | rd f64 %.3f f32 %.3f | rdg %.3f %.3f' % (d['ms_per_step'], d['reference_default']['f64']['ms_per_step'], d['reference_default']['f32']['ms_per_step'], d['reference_default_gmres']['f64']['ms_per_arnoldi_step'], d['reference_default_gmres']['f32']['ms_per_arnoldi_step']))
"
  done
done
