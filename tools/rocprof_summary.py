#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small summaries kept under profiles/.

  rocprof_summary.py stats <dir> <out.md> "<title>"     # --kernel-trace --stats run
  rocprof_summary.py pmc <fetch_dir> <write_dir> <out.json> [name-substring ...]
  rocprof_summary.py bygrid <dir> <out.md> "<title>"    # --kernel-trace run: launches grouped by (kernel, grid size)

PMC units follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, so it is
doubled; WRITE_SIZE is exact.  The two counters come from separate passes.
"""
import csv
import glob
import json
import os
import re
import sys

csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:70]


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit("no %s under %s" % (suffix, d))
    return hits[0]


def stats(d, out, title):
    rows = list(csv.DictReader(open(find(d, "_kernel_stats.csv"))))
    with open(out, "w") as fh:
        fh.write("# %s\n\n| kernel | calls | total ms | avg us | %% |\n|---|---|---|---|---|\n" % title)
        for r in rows:
            fh.write("| `%s` | %s | %.2f | %.1f | %s |\n" % (short(r["Name"]), r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))


def bygrid(d, out, title):
    """One template instance serves matrices of very different size (the AMG levels): group its launches by grid size."""
    acc = {}
    for r in csv.DictReader(open(find(d, "_kernel_trace.csv"))):
        key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
        a = acc.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    total = sum(a[1] for a in acc.values())
    with open(out, "w") as fh:
        fh.write("# %s\n\n| kernel | workgroups | calls | total ms | avg us | %% |\n|---|---|---|---|---|---|\n" % title)
        for (name, grid), (n, us) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            if us < 0.002 * total:
                continue
            fh.write("| `%s` | %d | %d | %.2f | %.1f | %.2f |\n" % (name, grid, n, us / 1e3, us / n, 100.0 * us / total))


def per_kernel(d, counter):
    acc = {}
    for r in csv.DictReader(open(find(d, "_counter_collection.csv"))):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        a = acc.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return acc


def pmc(fetch_dir, write_dir, out, filters):
    f, w = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) & set(w)):
        if filters and not any(s in k for s in filters):
            continue
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        nf, fs, _ = f[k]
        nw, ws, _ = w[k]
        rd = 2.0 * 1024.0 * fs / nf  # gfx950: FETCH_SIZE counts half
        wr = 1024.0 * ws / nw
        res[k] = {"launches": nf, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
    workload = {"elements_per_gpu": int(os.environ.get("FDD_PROFILE_ELEMENTS", "32")), "degree": int(os.environ.get("FDD_PROFILE_DEGREE", "7"))}
    json.dump({"workload": workload, "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950)", "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        print("%-60s n=%4d read %.1f MB write %.1f MB" % (k, v["launches"], v["read_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == "bygrid":
        bygrid(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5:])
