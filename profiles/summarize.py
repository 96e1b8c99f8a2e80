"""Turns the rocprofv3 outputs under gpurun_out/prof_* (profiles/collect.sh) into the committed summaries:
  profiles/<tag>_kernel_stats.csv   copy of rocprofv3 --kernel-trace --stats
  profiles/<tag>_summary.md         per-kernel mean duration + HBM traffic per launch
  profiles/traffic.json             per-stage HBM bytes per launch (read by bench.py's roofline.traffic)
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB, collected in separate
--pmc passes; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads, so it is doubled
(our gathers are 4-16 B per lane, so the doubled figure is an upper bound for them); WRITE_SIZE is taken as is."""
import csv
import json
import os
import re
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out")
STAGES = ["primary", "bounce", "extend", "shade", "connect", "resolve", "bloom", "tonemap"]


def template_args(name):
    m = re.search(r"<([^>]*)>", name)
    return [a.strip() for a in m.group(1).split(",")] if m else []


def stage_of(name):
    a = template_args(name)
    if len(a) >= 2 and a[1] == "true":
        return None  # traversal-counting variants (short pre-pass of bench.py): k_*<LDS, COUNT, ...>
    if "k_bounce" in name:
        return "primary" if len(a) >= 3 and a[2] == "true" else "bounce"  # k_bounce<LDS, COUNT, FIRST, VOL>
    for s in STAGES:
        if "k_" + s in name:
            return s
    return None


def counter_mean(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    if not os.path.exists(path):
        return {}
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        s = stage_of(row["Kernel_Name"])
        if s:
            acc[s][0] += float(row["Counter_Value"])
            acc[s][1] += 1
    return {s: v[0] / v[1] for s, v in acc.items() if v[1]}


stats_src = os.path.join(G, "prof_kt", "kt_kernel_stats.csv")
shutil.copy(stats_src, os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
dur = {}
for row in csv.DictReader(open(stats_src)):
    s = stage_of(row["Name"])
    if s:
        d = dur.setdefault(s, [0, 0.0])
        d[0] += int(row["Calls"])
        d[1] += float(row["TotalDurationNs"])
fetch = counter_mean(os.path.join(G, "prof_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
write = counter_mean(os.path.join(G, "prof_write", "write_counter_collection.csv"), "WRITE_SIZE")
traffic = {}
lines = ["# rocprofv3 summary (%s)" % tag, "",
         "Command: `rocprofv3 --kernel-trace --stats -- python bench.py --steps 16 --warmup 2 --no-cpu-baseline` (+ separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes at `--steps 4`).",
         "", "| stage | calls | mean duration (us) | FETCH_SIZE KiB/launch (raw) | WRITE_SIZE KiB/launch | HBM bytes/launch (2x fetch + write) |", "|---|---|---|---|---|---|"]
for s, (calls, tot) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    f, w = fetch.get(s), write.get(s)
    hbm = (2 * f + w) * 1024 if f is not None and w is not None else None
    traffic[s] = {"mean_duration_us": tot / calls / 1e3, "fetch_kib_raw": f, "write_kib": w, "hbm_bytes_per_launch": hbm}
    lines.append("| %s | %d | %.2f | %s | %s | %s |" % (s, calls, tot / calls / 1e3, "%.1f" % f if f is not None else "-", "%.1f" % w if w is not None else "-", "%.3e" % hbm if hbm else "-"))
open(os.path.join(ROOT, "profiles", tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print("\n".join(lines))
