#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/collect_profiles.sh (merged back under gpurun_out/) into the committed evidence:
  profiles/<tag>_kernel_stats.csv          rocprofv3 --kernel-trace --stats summary of `python bench.py --steps 10 --warmup 3`
  profiles/<tag>_bench.json                the JSON line that run printed
  profiles/dominant_kernel_traffic.json    HBM bytes per launch of bench.py's dominant kernel from the FETCH_SIZE / WRITE_SIZE passes
usage: pmc_traffic.py <tag>     (e.g. r01b)"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
line = [l for l in open(os.path.join(G, "prof_stats.log")) if l.startswith("{")][-1]
bench = json.loads(line)
kern = bench["roofline"]["kernel"]
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
shutil.copy(os.path.join(G, "prof_stats", "st_kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
json.dump(bench, open(os.path.join(ROOT, "profiles", f"{tag}_bench.json"), "w"), indent=1)


def per_launch(sub, prefix, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(os.path.join(G, sub, f"{prefix}_counter_collection.csv"))):
        if r["Counter_Name"] == counter and kern in r["Kernel_Name"]:
            tot += float(r["Counter_Value"])
            n += 1
    return tot / max(n, 1), n


fetch_kb, nf = per_launch("prof_fetch", "f", "FETCH_SIZE")
write_kb, nw = per_launch("prof_write", "w", "WRITE_SIZE")


def all_conv_kernels():
    """Per-launch HBM traffic of every conv kernel instantiation (bench.py's dominant kernel can differ between boxes when
    two instantiations have nearly the same total time)."""
    acc = {}
    for sub, prefix, counter, slot in (("prof_fetch", "f", "FETCH_SIZE", 0), ("prof_write", "w", "WRITE_SIZE", 1)):
        for r in csv.DictReader(open(os.path.join(G, sub, f"{prefix}_counter_collection.csv"))):
            name = r["Kernel_Name"]
            if r["Counter_Name"] != counter or "conv_mfma" not in name:
                continue
            key = name[name.index("conv_mfma"):name.index(">") + 1] if ">" in name else name
            e = acc.setdefault(key, [0.0, 0, 0.0, 0])
            e[slot * 2] += float(r["Counter_Value"])
            e[slot * 2 + 1] += 1
    return {k: (2.0 * v[0] / max(v[1], 1) + v[2] / max(v[3], 1)) * 1024.0 for k, v in acc.items()}

stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(G, "prof_stats", "st_kernel_stats.csv")))}
srow = next((v for k, v in stats.items() if kern in k), None)
out = {
    "kernel": kern,
    "traffic_bytes": (2.0 * fetch_kb + write_kb) * 1024.0,
    "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on `bench.py --steps 2 --warmup 1 --graph 0 "
           "--probe 0`; average over every dispatch of this kernel name; traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 "
           "FETCH_SIZE x2 correction of MI355X_MICROARCH.md (FETCH_SIZE/WRITE_SIZE are in KiB)",
    "fetch_size_kb_per_launch": fetch_kb, "write_size_kb_per_launch": write_kb, "dispatches_averaged": [nf, nw],
    "algorithmic_bytes_per_launch": bench["roofline"]["bytes_per_launch"],
    "bench_avg_us": bench["roofline"]["avg_us"],
    "rocprof_stats_avg_us": float(srow["AverageNs"]) / 1e3 if srow else None,
    "rocprof_stats_calls": int(srow["Calls"]) if srow else None,
    "round": tag,
    "traffic_bytes_by_kernel": all_conv_kernels(),
}
json.dump(out, open(os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
