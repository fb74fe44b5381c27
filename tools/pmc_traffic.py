#!/usr/bin/env python3
"""Turns the rocprofv3 output of tools/collect_profiles.sh (merged back under gpurun_out/) into the committed evidence:
  profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary of `python bench.py --steps 10 --warmup 3`
  profiles/<tag>_bench.json            the JSON line that run printed
  profiles/kernel_traffic.json         HBM bytes per launch of EVERY kernel of the step and per step by kernel family, from the
                                       FETCH_SIZE / WRITE_SIZE passes ((2*FETCH_SIZE + WRITE_SIZE)*1024: MI355X_MICROARCH.md)
  profiles/<tag>_mfma_lds_pmc.csv      per kernel: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD) and
                                       SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, averaged over its dispatches
usage: pmc_traffic.py <tag> [steps-in-the-pmc-runs: taken from the dispatch count of the once-per-step pack kernel]     (e.g. r03a)"""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
pmc_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3  # bench.py --steps 2 --warmup 1
line = [l for l in open(os.path.join(G, "prof_stats.log")) if l.startswith("{")][-1]
bench = json.loads(line)
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
shutil.copy(os.path.join(G, "prof_stats", "st_kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
json.dump(bench, open(os.path.join(ROOT, "profiles", f"{tag}_bench.json"), "w"), indent=1)


def short(name):
    """'void conv_mfma_pp_kernel<32, 4, 3, 1, 2>(ConvArgs, int)' -> 'conv_mfma_pp_kernel<32, 4, 3, 1, 2>'"""
    name = re.sub(r"^void ", "", name.strip())
    m = re.match(r"([A-Za-z_0-9:]+(?:<[^>]*>)?)", name)
    return m.group(1) if m else name


def counters(sub, prefix):
    acc = {}
    path = os.path.join(G, sub, f"{prefix}_counter_collection.csv")
    if not os.path.exists(path):
        return acc
    for r in csv.DictReader(open(path)):
        e = acc.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], [0.0, 0])
        e[0] += float(r["Counter_Value"])
        e[1] += 1
    return acc


fetch, write = counters("prof_fetch", "f"), counters("prof_write", "w")
# bench.py runs more steps than --steps + --warmup (loss-scale settling, the second loss mode): the weight pack is launched exactly
# once per step, so its dispatch count IS the number of steps a pass executed
for once in ("pack_weights_batched_kernel",):
    n_once = [v.get(c, [0.0, 0])[1] for v, c in ((fetch.get(once, {}), "FETCH_SIZE"), (write.get(once, {}), "WRITE_SIZE")) if v]
    if n_once:
        pmc_steps = max(n_once)
by_kernel, per_step, n_disp = {}, {}, {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, {}).get("FETCH_SIZE", [0.0, 1]), write.get(k, {}).get("WRITE_SIZE", [0.0, 1])
    by_kernel[k] = (2.0 * f[0] / max(f[1], 1) + w[0] / max(w[1], 1)) * 1024.0
    per_step[k] = (2.0 * f[0] + w[0]) * 1024.0 / pmc_steps
    n_disp[k] = max(f[1], w[1]) / pmc_steps


def family(k):
    if k.startswith(("bn_", "bn_act")):
        return "BN + activation"
    if k.startswith("conv_wgrad") or k.startswith("wgrad_reduce"):
        return "weight gradient (+ slab reduce)"
    if k.startswith("conv_mfma") or k.startswith("pack_"):
        return "conv forward + input gradient"
    if k.startswith(("maxpool", "upsample", "add_", "scalseq", "zoom", "copy_slice", "import_", "_Z19import", "warp_")):
        return "pooling / upsample / add / ScalSeq / image import"
    return "loss, optimizer, LDConv sampling, memsets, other"


fam = {}
for k, v in per_step.items():
    fam[family(k)] = fam.get(family(k), 0.0) + v
stats = {short(r["Name"]): r for r in csv.DictReader(open(os.path.join(G, "prof_stats", "st_kernel_stats.csv")))}
out = {
    "round": tag,
    "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on `bench.py --steps 2 --warmup 1 --graph 0 "
           "--probe 0`; per kernel name the average over every dispatch; traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 with the gfx950 "
           "FETCH_SIZE x2 correction of MI355X_MICROARCH.md (FETCH_SIZE/WRITE_SIZE are in KiB; Infinity-Cache hits are counted)",
    "step_traffic_bytes": sum(per_step.values()),
    "step_traffic_bytes_by_family": fam,
    "traffic_bytes_by_kernel": by_kernel,
    "step_bytes_by_kernel": per_step,
    "dispatches_per_step": n_disp,
    "rocprof_stats_avg_us": {k: float(v["AverageNs"]) / 1e3 for k, v in stats.items()},
    "bench_roofline_kernel": bench.get("roofline", {}).get("kernel") if bench.get("roofline") else None,
}
json.dump(out, open(os.path.join(ROOT, "profiles", "kernel_traffic.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("step_traffic_bytes", "step_traffic_bytes_by_family")}, indent=1))

mf, ld = counters("prof_mfma", "m"), counters("prof_lds", "l")
with open(os.path.join(ROOT, "profiles", f"{tag}_mfma_lds_pmc.csv"), "w") as fo:
    fo.write("kernel,dispatches,mfma_busy_cycles_per_dispatch,grbm_gui_active_per_xcd_per_dispatch,mfma_busy_frac_of_simd_cycles,"
             "lds_bank_conflict_cycles,lds_idx_active_cycles,lds_conflict_frac\n")
    for k in sorted(set(mf) | set(ld), key=lambda k: -per_step.get(k, 0.0)):
        m = mf.get(k, {})
        busy, gui = m.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0, 0]), m.get("GRBM_GUI_ACTIVE", [0.0, 0])
        n = max(busy[1], 1)
        # SQ_VALU_MFMA_BUSY_CYCLES sums the cycles each SIMD's matrix pipe is busy (16 per v_mfma_f32_16x16x32_f16) over the 1,024
        # SIMDs; GRBM_GUI_ACTIVE sums over the 8 XCDs -> busy fraction of all SIMD-cycles = fraction of the dense MFMA peak
        frac = busy[0] / (gui[0] / 8.0 * 1024.0) if gui[0] else 0.0
        l = ld.get(k, {})
        c, a = l.get("SQ_LDS_BANK_CONFLICT", [0.0, 0]), l.get("SQ_LDS_IDX_ACTIVE", [0.0, 0])
        fo.write(f"\"{k}\",{n},{busy[0] / n:.0f},{gui[0] / 8.0 / max(gui[1], 1):.0f},{frac:.4f},{c[0] / max(c[1], 1):.0f},{a[0] / max(a[1], 1):.0f},"
                 f"{(c[0] / a[0]) if a[0] else 0.0:.4f}\n")
print("wrote", f"profiles/{tag}_mfma_lds_pmc.csv")
# BASELINE.json configs[3] / configs[4]: the kernel stats of the LD training bench and of get_FPS.py (yolov8n-p2, 1280x1280, batch 32)
for sub, name in (("prof_ld", "ld"), ("prof_p2", "p2_1280")):
    src = os.path.join(G, sub, "st_kernel_stats.csv")
    if not os.path.exists(src):
        hits = [os.path.join(d, f) for d, _, fs in os.walk(os.path.join(G, sub)) for f in fs if f.endswith("kernel_stats.csv")]
        src = hits[0] if hits else None
    if src:
        shutil.copy(src, os.path.join(ROOT, "profiles", f"{tag}_{name}_kernel_stats.csv"))
        log = os.path.join(G, sub + ".log")
        if os.path.exists(log):
            keep = [l for l in open(log) if l.startswith("{") or l.startswith("model ")]
            if keep:
                open(os.path.join(ROOT, "profiles", f"{tag}_{name}_line.txt"), "w").write(keep[-1])
        print("wrote", f"profiles/{tag}_{name}_kernel_stats.csv")
