#!/bin/bash
# development A/B: the SELL row-length threshold (FDD_TUNE_CSR_SELL_MAX_ROW: matrices with more entries per row stay on the
# row-block kernel) on the box (AMG level 0: 7 entries per row, level 1: 27) and the Kershaw mesh (level 0: 15)
cd "$(dirname "$0")/.."
for mesh in box kershaw; do
  for sell in 32 16 12 32 16 12; do
    FDD_TUNE_CSR_SELL_MAX_ROW=$sell python3 bench.py --mesh $mesh --steps 6 --warmup 2 --no-time-to-tolerance --no-cpu-baseline --no-kershaw --no-stencil > gpurun_out/ab8.json 2>/dev/null
    python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/ab8.json') if l.startswith('{')][-1])
print('$mesh sell_max_row $sell: rd f64 %.3f f32 %.3f | rdg %.3f %.3f' % (d['reference_default']['f64']['ms_per_step'], d['reference_default']['f32']['ms_per_step'], d['reference_default_gmres']['f64']['ms_per_arnoldi_step'], d['reference_default_gmres']['f32']['ms_per_arnoldi_step']))
"
  done
done
