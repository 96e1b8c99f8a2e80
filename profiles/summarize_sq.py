"""SQ-side PMC summary of the path kernels (profiles/collect_pmc.sh) -> profiles/<tag>_pmc_sq.md.
VALU busy = SQ_ACTIVE_INST_VALU (quad-cycles) * 4 / (1024 SIMDs * kernel cycles); kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs."""
import collections
import csv
import re
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in ("prof_sq/sq_counter_collection.csv", "prof_sq2/sq2_counter_collection.csv"):
    for r in csv.DictReader(open(os.path.join(ROOT, "gpurun_out", f))):
        k = r["Kernel_Name"]
        a = [x.strip() for x in re.search(r"<([^>]*)>", k).group(1).split(",")] if "<" in k else []
        if len(a) >= 2 and a[1] == "true":
            continue
        name = None
        if "k_bounce" in k:
            name = "primary" if len(a) >= 3 and a[2] == "true" else "bounce"
        else:
            for s in ("extend", "shade", "connect", "resolve", "raygen"):
                if "k_" + s in k:
                    name = s
        if name:
            a = acc[name][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
lines = ["# SQ counters per launch (%s)" % tag, "",
         "`rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS` and",
         "`--pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE` on `python bench.py --steps 3 --warmup 1 --no-cpu-baseline`.", "",
         "| kernel | launches | VALU wave-instr | VMEM rd / wr | LDS instr | wave quad-cycles | waiting (WAIT_ANY) | kernel cycles (per XCD) | VALU busy |", "|---|---|---|---|---|---|---|---|---|"]
for name, d in acc.items():
    g = lambda c: d[c][0] / d[c][1] if c in d and d[c][1] else float("nan")
    cyc = g("GRBM_GUI_ACTIVE") / 8.0
    busy = g("SQ_ACTIVE_INST_VALU") * 4.0 / (1024.0 * cyc)
    lines.append("| %s | %d | %.3g | %.3g / %.3g | %.3g | %.3g | %.0f %% | %.3g | %.0f %% |" % (
        name, d["SQ_WAVES"][1], g("SQ_INSTS_VALU"), g("SQ_INSTS_VMEM_RD"), g("SQ_INSTS_VMEM_WR"), g("SQ_INSTS_LDS"), g("SQ_WAVE_CYCLES"),
        100.0 * g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), cyc, 100.0 * busy))
open(os.path.join(ROOT, "profiles", tag + "_pmc_sq.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
