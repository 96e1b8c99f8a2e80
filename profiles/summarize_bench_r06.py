"""gpurun_out/prof_bench/ (profiles/collect_bench_r06.sh: the headline Cornell bench under rocprofv3) ->
  profiles/r06_cornell_kernel_stats.csv, profiles/r06_cornell_summary.md, and profiles/traffic.json, which bench.py reads for
  roofline.traffic / valu_busy.  traffic.json also takes the atrium / glass-bust stage figures from profiles/r06_<scene>_p2_counters.json
  (profiles/summarize_r06.py) when they exist.
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (KiB units; FETCH doubled per the gfx950 note in MI355X_MICROARCH.md
section HBM — an upper bound for 16-byte gathers); VALU busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8);
lanes = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (mean active lanes per VALU instruction, of 64; a fully active kernel such as raygen reads 64)."""
import collections, csv, glob, importlib, json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SOURCE_ID = importlib.import_module("vulkan-path-tracer_amd._build").source_id()   # the build these counters belong to (bench.py refuses entries of another one)
G = os.path.join(ROOT, "gpurun_out", "prof_bench")

def stage(name):
    m = re.search(r"<([^>]*)>", name)
    a = [x.strip() for x in m.group(1).split(",")] if m else []
    if "k_whole" in name:   # the whole-path launch (round 4): bench.py times it under "primary"; <COUNT, STRICT, PLAIN>
        return None if a and a[0] == "true" else "primary"
    if "k_bounce" in name:
        if len(a) >= 2 and a[1] == "true":
            return None   # traversal-counting variants (bench.py's counting pass)
        return "primary" if len(a) >= 3 and a[2] == "true" else "bounce"
    return "resolve" if "k_resolve" in name else None

ks = glob.glob(os.path.join(G, "kt", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(ks, os.path.join(ROOT, "profiles", "r06_cornell_kernel_stats.csv"))
dur = {}
for r in csv.DictReader(open(ks)):
    s = stage(r["Name"])
    if s:
        d = dur.setdefault(s, [0, 0.0]); d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(G, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        s = stage(r["Kernel_Name"])
        if s:
            acc[s][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[s][r["Counter_Name"]] += 1
out = {"_source_id": SOURCE_ID, "cornell_1080p_d8": {}}
knames = collections.defaultdict(set)
for r in csv.DictReader(open(ks)):
    if stage(r["Name"]): knames[stage(r["Name"])].add(r["Name"].split("(")[0])
lines = ["# Cornell 1080p depth 8 (bench.py headline workload; `primary` is the whole-path launch k_whole where AUTO takes it, else k_bounce<FIRST>) — rocprofv3 summary (r06)", "",
         "`python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-workloads` under profiles/collect_bench_r06.sh (kernel trace + separate PMC passes).", "",
         "| stage | calls | mean us | HBM bytes / launch (2 x FETCH + WRITE) | fetched | written | VALU busy (corrected: min(raw, 100 %)) | lanes / VALU instr | wait | L2 hit |", "|---|---|---|---|---|---|---|---|---|---|"]
flops_rows = []
for s, (calls, tot) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    a, c = acc[s], cnt[s]
    per = lambda k: a[k] / c[k] if c[k] else float("nan")
    fetch, write = per("FETCH_SIZE") * 1024, per("WRITE_SIZE") * 1024
    busy = a["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * a["GRBM_GUI_ACTIVE"] / 8) if a["GRBM_GUI_ACTIVE"] else float("nan")
    # GRBM_GUI_ACTIVE comes from another pass with the same launches: scale by launch counts
    if c["GRBM_GUI_ACTIVE"] and c["SQ_ACTIVE_INST_VALU"]:
        busy = (a["SQ_ACTIVE_INST_VALU"] / c["SQ_ACTIVE_INST_VALU"]) * 4 / (1024 * (a["GRBM_GUI_ACTIVE"] / c["GRBM_GUI_ACTIVE"]) / 8)
    lanes = a["SQ_THREAD_CYCLES_VALU"] / a["SQ_ACTIVE_INST_VALU"] if a["SQ_ACTIVE_INST_VALU"] else float("nan")
    wait = a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"] if a["SQ_WAVE_CYCLES"] else float("nan")
    l2 = a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"]) if a["TCC_HIT_sum"] + a["TCC_MISS_sum"] else float("nan")
    # round 4: the fp32 operation mix (separate `flops` pass): FLOP = (ADD + MUL + TRANS + 2 x FMA) wave-instructions x 64 lanes x lane use
    fp_instr = a["SQ_INSTS_VALU_ADD_F32"] + a["SQ_INSTS_VALU_MUL_F32"] + a["SQ_INSTS_VALU_FMA_F32"] + a["SQ_INSTS_VALU_TRANS_F32"]
    nfl = max(c["SQ_INSTS_VALU_FMA_F32"], 1)
    flop_per_launch = (a["SQ_INSTS_VALU_ADD_F32"] + a["SQ_INSTS_VALU_MUL_F32"] + a["SQ_INSTS_VALU_TRANS_F32"] + 2 * a["SQ_INSTS_VALU_FMA_F32"]) / nfl * 64 * (lanes / 64.0)
    tflops = flop_per_launch / (tot / calls * 1e-9) / 1e12 if fp_instr else float("nan")
    fp_share = fp_instr / a["SQ_INSTS_VALU"] * (c["SQ_INSTS_VALU"] / nfl) if a["SQ_INSTS_VALU"] and fp_instr else float("nan")   # (SQ_INSTS_VALU is collected in two passes)
    flops_rows.append("| %s | %.0f %% | %.2f | %.3f |" % (s, 100 * fp_share, tflops, tflops / 157.3))
    out["cornell_1080p_d8"][s] = {"fp32_tflops": None if tflops != tflops else round(tflops, 2), "fp32_share_of_valu_instr": None if fp_share != fp_share else round(fp_share, 3),
                                  "mean_duration_us": tot / calls / 1e3, "hbm_bytes_per_launch": 2 * fetch + write, "fetch_bytes_raw": fetch, "write_bytes": write,
                                  "valu_busy": round(min(busy, 1.0), 3), "valu_busy_raw": round(busy, 3), "lanes_per_valu_instr": round(lanes, 1), "wait": round(wait, 3), "l2_hit": round(l2, 3),
                                  "kernel_names": sorted(knames[s])}
    lines.append("| %s | %d | %.1f | %.3e | %.3e | %.3e | %.0f %% | %.1f | %.0f %% | %.0f %% |" % (s, calls, tot / calls / 1e3, 2 * fetch + write, 2 * fetch, write, 100 * min(busy, 1.0), lanes, 100 * wait, 100 * l2))
for scene, wl in (("atrium_p2", "atrium_1080p_d8"), ("bust_p2", "glass_bust_1080p_d32"), ("atrium_p2_3840x2160", "atrium_4k_d8")):
    p = os.path.join(ROOT, "profiles", "r06_%s_counters.json" % scene)   # profiles/summarize_r06.py; only this round's passes count
    if os.path.exists(p):
        d = json.load(open(p)); o = out.setdefault(wl, {})
        names = {"trace_vote": "extend", "shade_stream": "shade", "join": "join", "shadow_sky": "shadow_sky", "shadow_light": "shadow_light", "finish": "bounce", "refill_stream": "primary_refill"}
        for k, v in d.items():
            if k in names and v["launches"]:
                o[names[k]] = {"mean_duration_us": v["total_ms"] * 1e3 / v["launches"], "hbm_bytes_per_launch": (v["fetched_GB"] + v["written_GB"]) * 1e9 / v["launches"],
                               "valu_busy": round(v["valu_busy"], 3), "valu_busy_raw": round(v["valu_busy_raw"], 3), "lanes_per_valu_instr": round(v["lane_use"] * 64, 1), "wait": round(v["wait"], 3), "l2_hit": round(v["l2_hit"], 3),
                               "kernel_names": v.get("kernel_names", [])}
        if "shadow_sky" in o and "shadow_light" in o:   # bench.py times both launches under one stage name
            a, b = o["shadow_sky"], o["shadow_light"]
            o["shadow"] = {"mean_duration_us": (a["mean_duration_us"] + b["mean_duration_us"]) / 2, "hbm_bytes_per_launch": (a["hbm_bytes_per_launch"] + b["hbm_bytes_per_launch"]) / 2,
                           "valu_busy": round((a["valu_busy"] * a["mean_duration_us"] + b["valu_busy"] * b["mean_duration_us"]) / (a["mean_duration_us"] + b["mean_duration_us"]), 3),
                           "lanes_per_valu_instr": round((a["lanes_per_valu_instr"] * a["mean_duration_us"] + b["lanes_per_valu_instr"] * b["mean_duration_us"]) / (a["mean_duration_us"] + b["mean_duration_us"]), 1),
                           "kernel_names": sorted(set(a["kernel_names"] + b["kernel_names"]))}
lds_rows = []
for s_, (calls, tot) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    a, c = acc[s_], cnt[s_]
    if not c["SQ_INSTS_LDS"]:
        continue
    per = lambda k: a[k] / c[k] if c[k] else float("nan")
    # SQ_ACTIVE_INST_LDS / SQ_LDS_* count quad-cycles summed over the SIMDs' wave slots (as SQ_ACTIVE_INST_VALU does): x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) = share of time
    gui = per("GRBM_GUI_ACTIVE")
    share = lambda k: per(k) * 4 / (1024 * gui / 8) if gui else float("nan")
    lds_rows.append("| %s | %.3g | %.0f %% | %.0f %% | %.0f %% | %.1f %% |" % (s_, per("SQ_INSTS_LDS"), 100 * share("SQ_ACTIVE_INST_LDS"), 100 * share("SQ_LDS_IDX_ACTIVE"), 100 * share("SQ_WAIT_INST_LDS"),
                                                                      100 * per("SQ_LDS_BANK_CONFLICT") / max(per("SQ_LDS_IDX_ACTIVE"), 1.0)))
    out["cornell_1080p_d8"][s_].update({"lds_instr_per_launch": per("SQ_INSTS_LDS"), "lds_active_share": round(share("SQ_ACTIVE_INST_LDS"), 3), "lds_idx_active_share": round(share("SQ_LDS_IDX_ACTIVE"), 3),
                                        "lds_wait_share": round(share("SQ_WAIT_INST_LDS"), 3), "lds_bank_conflict_of_active": round(per("SQ_LDS_BANK_CONFLICT") / max(per("SQ_LDS_IDX_ACTIVE"), 1.0), 4)})
if lds_rows:
    lines += ["", "LDS side (`lds` pass; shares are of the launch's time, normalised as VALU busy is; bank conflicts as a share of the cycles the LDS index unit is active):", "",
              "| stage | LDS wave-instructions / launch | LDS instruction active | LDS index unit active | waiting on an LDS instruction | bank-conflict cycles of active |", "|---|---|---|---|---|---|"] + lds_rows
lines += ["", "fp32 operation mix (`flops` pass: SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32; FLOP = (ADD + MUL + TRANS + 2 x FMA) wave-instructions x 64 lanes x lane use; peak 157.3 TFLOP/s assumes a packed FMA on every lane every cycle):", "",
          "| stage | fp32 share of VALU instructions | fp32 TFLOP/s | of 157.3 |", "|---|---|---|---|"] + flops_rows
open(os.path.join(ROOT, "profiles", "r06_cornell_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print("\n".join(lines))
