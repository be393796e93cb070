#!/bin/bash
# development A/B of one environment knob on the headline and the reference-default steps: tools/ab_env.sh NAME v1 v2 ... [-- bench args]
cd "$(dirname "$0")/.."
name=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" = "--" ] && shift
for v in "${vals[@]}"; do
  env $name=$v python3 bench.py --steps 6 --warmup 2 --no-time-to-tolerance --no-cpu-baseline --no-kershaw --no-stencil "$@" > gpurun_out/ab_env.json 2>/dev/null
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/ab_env.json') if l.startswith('{')][-1])
print('$name=$v: headline %.3f | rd f64 %.3f f32 %.3f | rdg %.3f %.3f' % (d['ms_per_step'], d['reference_default']['f64']['ms_per_step'], d['reference_default']['f32']['ms_per_step'], d['reference_default_gmres']['f64']['ms_per_arnoldi_step'], d['reference_default_gmres']['f32']['ms_per_arnoldi_step']))
"
done
