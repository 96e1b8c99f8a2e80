"""gpurun_out/prof_post_<size>_<schedule>/ (profiles/collect_post_r03.sh) -> profiles/r03_post_<size>_summary.md (+ the kernel stats csv).
Per kernel of the post chain: launches and mean duration (rocprofv3 --kernel-trace --stats), HBM-side bytes per call (FETCH_SIZE doubled as
in the other r02 / r03 summaries, WRITE_SIZE, both KiB), then per schedule the sum and SURVEY 8d's algorithmic 143 B / pixel over it."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
size = sys.argv[1]
w, h = (3840, 2160) if size == "4k" else (1920, 1080)
lines = ["# Post chain at %dx%d — rocprofv3 summary (r03)" % (w, h), "",
         "`SIZE=%s SCHEDULE=<s> python tests/gpu_post_run.py` under profiles/collect_post_r03.sh: 1 warm-up + 20 measured `vpt_postprocess` calls per schedule." % size,
         "SURVEY 8d's algorithmic traffic of the chain as the reference schedules it: 143 B per full-resolution pixel = %.0f MB." % (143 * w * h / 1e6), ""]
for sched in ("reference_passes", "fused"):
    G = os.path.join(ROOT, "gpurun_out", "prof_post_%s_%s" % (size, sched))
    ks = glob.glob(os.path.join(G, "kt", "**", "*kernel_stats.csv"), recursive=True)
    if not ks:
        continue
    shutil.copy(ks[0], os.path.join(ROOT, "profiles", "r03_post_%s_%s_kernel_stats.csv" % (size, sched)))
    ev = json.loads(open(os.path.join(G, "events.json")).read().strip().split("\n")[-1])
    calls = ev["reps"] + 1
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(os.path.join(G, "*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
    lines += ["## schedule: %s" % sched, "", "| kernel | launches per call | mean us | us per call | HBM-side MB per call (2 x FETCH + WRITE) |", "|---|---|---|---|---|"]
    tot_us = tot_mb = 0.0
    for r in csv.DictReader(open(ks[0])):
        name = r["Name"].split("(")[0]
        if not any(k in name for k in ("k_bloom", "k_tonemap", "k_post_final")):
            continue
        n, mean = int(r["Calls"]), float(r["AverageNs"]) / 1e3
        mb = (2 * acc[name]["FETCH_SIZE"] + acc[name]["WRITE_SIZE"]) * 1024 / 1e6 / calls
        tot_us += n * mean / calls; tot_mb += mb
        lines.append("| `%s` | %.1f | %.2f | %.2f | %.1f |" % (name.replace("vpt::", ""), n / calls, mean, n * mean / calls, mb))
    e = ev[sched]
    lines += ["", "Sum of kernel durations per call: **%.1f us** (rocprofv3) / %.1f us (the library's HIP events around the same launches); %d launches per call; HBM-side traffic %.0f MB per call."
              % (tot_us, e["gpu_ms"] * 1e3, e["launches"], tot_mb),
              "SURVEY 8d fraction: %.0f MB / %.1f us = %.2f TB/s = **%.2f of the 8 TB/s peak**; physical: %.0f MB / %.1f us = %.2f TB/s = %.2f.  One call incl. the blocking sync and the 8 / 33 MB RGBA8 read-back to the host: %.2f ms."
              % (143 * w * h / 1e6, tot_us, 143 * w * h / tot_us / 1e6, 143 * w * h / tot_us / 1e6 / 8, tot_mb, tot_us, tot_mb / tot_us, tot_mb / tot_us / 8, e["call_wall_ms_incl_readback"]), ""]
open(os.path.join(ROOT, "profiles", "r03_post_%s_summary.md" % size), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
