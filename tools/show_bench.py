#!/usr/bin/env python3
"""Pretty-print the JSON line bench.py wrote (last line of the given file)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"value {d['value']:.4e} {d['unit']}   ms/step {d['ms_per_step']:.3f}   n_gpus {d['n_gpus']}   last residual {d['last_residual_norm']:.3e}   setup {d['setup_s']:.1f} s")
r = d["roofline"]
if r:
    print(f"roofline: {r['kernel']}  {r['achieved']:.0f} GB/s = {100 * r['frac']:.1f}% of {r['peak']:.0f}  ({r['launches']} launches, avg {r['avg_launch_us']:.1f} us)")
tot = 0.0
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["total_ms"]):
    tot += v["total_ms"]
    print(f"  {k:40s} n={v['launches']:4d} avg_us={v['avg_us']:8.1f} total_ms={v['total_ms']:7.2f} GB/s={v['GBps']:7.0f}")
print(f"  instrumented total {tot:.2f} ms of {d['ms_per_step'] * d['steps']:.2f} ms timed")
if d.get("cpu_baseline"):
    print("cpu_baseline:", d["cpu_baseline"])
