"""gpurun_out/prof_<scene>_p<pipe>/ (profiles/collect_r03.sh) -> profiles/<TAG, default r03>_<scene>_p<pipe>_kernel_stats.csv + _summary.md.
SQ counters are quad-cycle based; wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES; VALU busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs x
GRBM_GUI_ACTIVE / 8) as in r01; lane use = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): mean fraction of the 64 lanes active per VALU instruction;
FETCH_SIZE / WRITE_SIZE in KiB, FETCH doubled (gfx950 note in MI355X_MICROARCH.md)."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene, pipe = sys.argv[1], sys.argv[2]
tag = "%s_%s_p%s" % (os.environ.get("TAG", "r03"), scene, pipe)
G = os.path.join(ROOT, "gpurun_out", "prof_%s_p%s" % (scene, pipe))
STAGES = ("raygen", "extend", "trace_vote", "trace_shadow", "shade_stream", "shade", "connect", "join", "resolve", "bounce", "prepare")
def stage(k):
    for s in STAGES:
        if "k_" + s in k:
            if s == "trace_shadow":
                return "shadow_light" if "ILb1E" in k or "<true" in k else "shadow_sky"
            return s
    return None
ks = glob.glob(os.path.join(G, "kt", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(ks, os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
dur = {}
for r in csv.DictReader(open(ks)):
    s = stage(r["Name"])
    if s:
        d = dur.setdefault(s, [0, 0.0]); d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(os.path.join(G, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        s = stage(r["Kernel_Name"])
        if s:
            acc[s][r["Counter_Name"]] += float(r["Counter_Value"])
rows = {}
lines = ["# %s, pipeline %s — rocprofv3 summary (%s)" % (scene, pipe, os.environ.get("TAG", "r03")), "",
         "`SCENE=%s PIPE=%s python tests/gpu_atrium_run.py` under profiles/collect_r03.sh: 1920x1080, %s frames in flight, 2 measured batches (+1 warm-up, included in the sums)." % (scene, pipe, os.environ.get("FRAMES", "64")), "",
         "| stage | launches | total ms | wait | VALU busy | lane use | VALU wave-instr | L2 hit | fetched GB (2 x FETCH_SIZE) | written GB |", "|---|---|---|---|---|---|---|---|---|---|"]
for s, (calls, tot) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    a = acc[s]
    div = lambda x, y: x / y if y else float("nan")
    wait = div(a["SQ_WAIT_ANY"], a["SQ_WAVE_CYCLES"])
    l2 = div(a["TCC_HIT_sum"], a["TCC_HIT_sum"] + a["TCC_MISS_sum"])
    busy = div(a["SQ_ACTIVE_INST_VALU"] * 4, 1024 * a["GRBM_GUI_ACTIVE"] / 8)
    lane = div(a["SQ_THREAD_CYCLES_VALU"], 64 * a["SQ_ACTIVE_INST_VALU"])
    rows[s] = {"launches": calls, "total_ms": tot / 1e6, "wait": wait, "valu_busy": busy, "lane_use": lane, "valu_wave_instr": a["SQ_INSTS_VALU"], "l2_hit": l2,
               "fetched_GB": 2 * a["FETCH_SIZE"] * 1024 / 1e9, "written_GB": a["WRITE_SIZE"] * 1024 / 1e9, "raw": dict(a)}
    lines.append("| %s | %d | %.2f | %.0f %% | %.0f %% | %.0f %% | %.3g | %.0f %% | %.2f | %.2f |" % (s, calls, tot / 1e6, 100 * wait, 100 * busy, 100 * lane, a["SQ_INSTS_VALU"], 100 * l2, rows[s]["fetched_GB"], rows[s]["written_GB"]))
open(os.path.join(ROOT, "profiles", tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(rows, open(os.path.join(ROOT, "profiles", tag + "_counters.json"), "w"), indent=1)
print("\n".join(lines))
