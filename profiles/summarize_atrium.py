"""Atrium (285k triangles, staged pipeline) profile -> profiles/<tag>_atrium_kernel_stats.csv + <tag>_atrium_summary.md.
Inputs: gpurun_out/prof_a0 (kernel-trace stats), prof_a1 (SQ), prof_a2 (TCC hit/miss), prof_a3 (FETCH_SIZE) from
profiles/collect_atrium.sh on `python tests/gpu_atrium_run.py` (1080p, depth 8, 8 frames in flight, 2 batches)."""
import collections
import csv
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out")
shutil.copy(os.path.join(G, "prof_a0", "a0_kernel_stats.csv"), os.path.join(ROOT, "profiles", tag + "_atrium_kernel_stats.csv"))


def stage(k):
    for s in ("raygen", "extend", "shade", "connect", "resolve"):
        if "k_" + s in k:
            return s
    return None


dur = {}
for r in csv.DictReader(open(os.path.join(G, "prof_a0", "a0_kernel_stats.csv"))):
    s = stage(r["Name"])
    if s:
        d = dur.setdefault(s, [0, 0.0]); d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in ("prof_a1/a1_counter_collection.csv", "prof_a2/a2_counter_collection.csv", "prof_a3/a3_counter_collection.csv"):
    p = os.path.join(G, f)
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        s = stage(r["Kernel_Name"])
        if s:
            acc[s][r["Counter_Name"]] += float(r["Counter_Value"])
lines = ["# Atrium (285k triangles, staged pipeline) — rocprofv3 summary (%s)" % tag, "",
         "`python tests/gpu_atrium_run.py`: 1920x1080, depth 8, 8 frames in flight, 2 batches of 8 frames (16.6M samples each); sums over both batches.",
         "SQ counters are quad-cycles; wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES; VALU busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * GRBM_GUI_ACTIVE / 8);",
         "L2 hit = TCC_HIT / (TCC_HIT + TCC_MISS); FETCH_SIZE in KiB, doubled for gfx950 (upper bound for 16 B gathers).", "",
         "| stage | launches | total ms | waves | wait | VALU busy | VALU wave-instr | VMEM rd instr | L2 hit | fetched GB (2 x FETCH_SIZE) |", "|---|---|---|---|---|---|---|---|---|---|"]
for s, (calls, tot) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    a = acc[s]
    wait = a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"] if a["SQ_WAVE_CYCLES"] else float("nan")
    l2 = a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"]) if a["TCC_HIT_sum"] + a["TCC_MISS_sum"] else float("nan")
    busy = a["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * a["GRBM_GUI_ACTIVE"] / 8) if a["GRBM_GUI_ACTIVE"] else float("nan")
    lines.append("| %s | %d | %.2f | %.3g | %.0f %% | %.0f %% | %.3g | %.3g | %.0f %% | %.2f |" % (
        s, calls, tot / 1e6, a["SQ_WAVES"], 100 * wait, 100 * busy, a["SQ_INSTS_VALU"], a["SQ_INSTS_VMEM_RD"], 100 * l2, 2 * a["FETCH_SIZE"] * 1024 / 1e9))
open(os.path.join(ROOT, "profiles", tag + "_atrium_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
