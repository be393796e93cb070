#!/usr/bin/env python3
"""Pretty-print the JSON line bench.py wrote (last line of the given file)."""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"value {d['value']:.4e} {d['unit']}   ms/step {d['ms_per_step']:.3f}   n_gpus {d['n_gpus']}   last residual {d['last_residual_norm']:.3e}   setup {d['setup_s']:.1f} s")
r = d["roofline"]
if r:
    print(f"roofline: {r['kernel']}  {r['achieved']:.0f} GB/s = {100 * r['frac']:.1f}% of {r['peak']:.0f}  ({r['launches']} launches, avg {r['avg_launch_us']:.1f} us)")
for k, v in (d.get("spmv") or {}).items():
    formula = f"  (CSR byte formula: {v['formula_GBps']:.0f} GB/s)" if "formula_GBps" in v else ""
    print(f"spmv {k}: {v['avg_us']:.1f} us  {v['GBps_moved']:.0f} GB/s moved = {100 * v['frac_moved_of_hbm_peak']:.1f}% of HBM peak{formula}")
if r and r.get("traffic"):
    print(f"  PMC traffic {r['traffic'] / 1e6:.1f} MB per launch vs algorithmic {r['algorithmic_bytes_per_launch'] / 1e6:.1f} MB")
tot = 0.0
for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["total_ms"]):
    tot += v["total_ms"]
    print(f"  {k:40s} n={v['launches']:4d} avg_us={v['avg_us']:8.1f} total_ms={v['total_ms']:7.2f} GB/s={v['GBps']:7.0f}")
print(f"  instrumented total {tot:.2f} ms of {d['ms_per_step'] * d['steps']:.2f} ms timed")
for key in ("reference_default", "reference_default_gmres"):
    o = d.get(key) or {}
    for prec in ("f64", "f32"):
        if prec in o:
            ms = o[prec].get("ms_per_step", o[prec].get("ms_per_arnoldi_step"))
            print(f"{key}.{prec}: {ms:.3f} ms per step, to 1e-7: {o[prec].get('to_1e-7')}")
k = d.get("kershaw") or {}
if k.get("reference_default"):
    kr = k["reference_default"]
    print(f"kershaw reference_default: {kr['ms_per_step']:.3f} ms per step (f32 {kr.get('f32_ms_per_step', float('nan')):.3f}), pcg {kr.get('to_1e-7')}, gmres {kr.get('gmres_to_1e-7')}")
if d.get("cpu_baseline"):
    print("cpu_baseline:", d["cpu_baseline"])
