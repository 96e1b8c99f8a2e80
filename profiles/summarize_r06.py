"""gpurun_out/prof_<scene>_p<pipe>[_WxH]/ (profiles/collect_r06.sh) -> profiles/<TAG, default r06>_<scene>_p<pipe>[_WxH]_kernel_stats.csv + _summary.md + _counters.json.
VALU busy is printed CORRECTED: the raw quotient reads up to 1.2 on kernels that saturate VALU issue (profiles/r02_valu_calibration.md: 0.93-1.0 is saturation), so the
tables show min(raw, 100 %) and the JSON keeps both (valu_busy, valu_busy_raw).  Every stage lists the kernel names it was collected from (kernel_names).
SQ counters are quad-cycle based; wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES; VALU busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs x
GRBM_GUI_ACTIVE / 8) as in r01; lane use = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): mean fraction of the 64 lanes active per VALU instruction;
FETCH_SIZE / WRITE_SIZE in KiB, FETCH doubled (gfx950 note in MI355X_MICROARCH.md)."""
import collections, csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene, pipe = sys.argv[1], sys.argv[2]
size = sys.argv[3] if len(sys.argv) > 3 else ""
tag = "%s_%s_p%s%s" % (os.environ.get("TAG", "r06"), scene, pipe, "_" + size if size else "")
G = os.path.join(ROOT, "gpurun_out", "prof_%s_p%s%s" % (scene, pipe, "_" + size if size else ""))
STAGES = ("raygen", "refill_stream", "extend", "trace_vote", "trace_shadow", "shade_stream", "shade", "connect", "join", "resolve", "finish", "bounce", "prepare")
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
knames = collections.defaultdict(list)
for r in csv.DictReader(open(ks)):
    s = stage(r["Name"])
    if s:
        d = dur.setdefault(s, [0, 0.0]); d[0] += int(r["Calls"]); d[1] += float(r["TotalDurationNs"])
        if "k_finish_done" not in r["Name"]: knames[s].append(r["Name"].split("(")[0])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
passes = collections.defaultdict(set)   # a counter collected in several passes (GRBM_GUI_ACTIVE, SQ_INSTS_VALU, ...) is averaged over them
for f in glob.glob(os.path.join(G, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        s = stage(r["Kernel_Name"])
        if s:
            acc[s][r["Counter_Name"]] += float(r["Counter_Value"]); passes[r["Counter_Name"]].add(f)
for s in acc:
    for k in acc[s]:
        acc[s][k] /= max(len(passes[k]), 1)
rows = {}
lines = ["# %s, pipeline %s — rocprofv3 summary (%s)" % (scene, pipe, os.environ.get("TAG", "r06")), "",
         "`SCENE=%s PIPE=%s python tests/gpu_atrium_run.py` under profiles/collect_r06.sh: %s, FRAMES=%s (0: the library's own schedule — batches of 904 frames with 113 frames of paths resident at 1080p, what bench.py's workloads run), 2 measured batches (+1 warm-up, included in the sums)." % (scene, pipe, size or "1920x1080", os.environ.get("FRAMES", "0")), "",
         "| stage | launches | total ms | mean ms | wait | VALU busy (corrected) | lane use | VALU wave-instr | L2 hit | fetched GB (2 x FETCH_SIZE) | written GB | GB per launch | TB/s |", "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for s, (calls, tot) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
    a = acc[s]
    div = lambda x, y: x / y if y else float("nan")
    wait = div(a["SQ_WAIT_ANY"], a["SQ_WAVE_CYCLES"])
    l2 = div(a["TCC_HIT_sum"], a["TCC_HIT_sum"] + a["TCC_MISS_sum"])
    busy = div(a["SQ_ACTIVE_INST_VALU"] * 4, 1024 * a["GRBM_GUI_ACTIVE"] / 8)
    lane = div(a["SQ_THREAD_CYCLES_VALU"], 64 * a["SQ_ACTIVE_INST_VALU"])
    rows[s] = {"launches": calls, "total_ms": tot / 1e6, "wait": wait, "valu_busy": min(busy, 1.0) if busy == busy else busy, "valu_busy_raw": busy, "lane_use": lane, "valu_wave_instr": a["SQ_INSTS_VALU"], "l2_hit": l2,
               "fetched_GB": 2 * a["FETCH_SIZE"] * 1024 / 1e9, "written_GB": a["WRITE_SIZE"] * 1024 / 1e9, "kernel_names": sorted(set(knames[s])), "raw": dict(a)}
    gb = (rows[s]["fetched_GB"] + rows[s]["written_GB"]) / max(calls, 1)
    lines.append("| %s | %d | %.2f | %.3f | %.0f %% | %.0f %% | %.0f %% | %.3g | %.0f %% | %.2f | %.2f | %.3f | %.2f |" % (s, calls, tot / 1e6, tot / 1e6 / max(calls, 1), 100 * wait, 100 * min(busy, 1.0), 100 * lane, a["SQ_INSTS_VALU"], 100 * l2,
                 rows[s]["fetched_GB"], rows[s]["written_GB"], gb, (rows[s]["fetched_GB"] + rows[s]["written_GB"]) / max(tot / 1e9, 1e-12) / 1e3))
# ---- round 3 additions: vector-memory pipeline, fp32 mix, instruction cache (only when those passes were collected)
if any("TA_TA_BUSY_sum" in acc[s] for s in acc):
    lines += ["", "Vector-memory pipeline, fp32 operation mix and instruction fetch (separate passes: `ta`, `flops`, `icache`).  TA busy = TA_TA_BUSY_sum / (32 x GRBM_GUI_ACTIVE): GRBM_GUI_ACTIVE sums the 8 XCDs, each with 32 CUs / TAs (TCP_GATE_EN1_sum / GRBM_GUI_ACTIVE reads 31.7 on a kernel that keeps every CU's L1 clocked);",
              "L1 accesses per VMEM-read wave-instruction = TCP_TOTAL_CACHE_ACCESSES_sum / SQ_INSTS_VMEM_RD (64 = every lane its own 64-byte request, 16 = four lanes per request);",
              "fp32 FLOP = (ADD + MUL + TRANS + 2 x FMA) wave-instructions x 64 lanes x lane use; peak 157.3 TFLOP/s (which assumes packed FMA on every lane).", "",
              "L1 accesses per clock per CU = TCP_TOTAL_CACHE_ACCESSES_sum / (256 CUs x kernel time x 2.4 GHz): a CU's L1 takes about one access per clock (tests/tools/gather_calib.hip).", "",
              "| stage | TA busy | L1 accesses / clk / CU | L1 accesses / VMEM-rd instr | TCP pending-stall share | fp32 share of VALU instr | fp32 TFLOP/s | of 157.3 | I-cache hit | SALU / VALU instr |", "|---|---|---|---|---|---|---|---|---|---|"]
    for s, (calls, tot) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        a = acc[s]
        div = lambda x, y: x / y if y else float("nan")
        have_flops = "SQ_INSTS_VALU_FMA_F32" in a     # (before the lookups below: `a` is a defaultdict)  the fp32-mix / instruction-cache passes are collected for the headline bench only: say so instead of printing zeros
        have_ic = "SQC_ICACHE_REQ" in a
        fp = a["SQ_INSTS_VALU_ADD_F32"] + a["SQ_INSTS_VALU_MUL_F32"] + a["SQ_INSTS_VALU_FMA_F32"] + a["SQ_INSTS_VALU_TRANS_F32"]
        flop = (a["SQ_INSTS_VALU_ADD_F32"] + a["SQ_INSTS_VALU_MUL_F32"] + a["SQ_INSTS_VALU_TRANS_F32"] + 2 * a["SQ_INSTS_VALU_FMA_F32"]) * 64 * rows[s]["lane_use"]
        tfl = div(flop, tot / 1e9) / 1e12
        rows[s].update({"ta_busy": div(a["TA_TA_BUSY_sum"], 32 * a["GRBM_GUI_ACTIVE"]), "fp32_tflops": tfl,
                        "l1_accesses_per_vmem_rd": div(a["TCP_TOTAL_CACHE_ACCESSES_sum"], a["SQ_INSTS_VMEM_RD"]), "icache_hit": div(a["SQC_ICACHE_HITS"], a["SQC_ICACHE_REQ"])})
        rows[s]["l1_accesses_per_clk_per_cu"] = div(a["TCP_TOTAL_CACHE_ACCESSES_sum"], 256 * (tot / 1e9) * 2.4e9)
        nc = "not collected"
        lines.append("| %s | %.0f %% | %.2f | %.1f | %.0f %% | %s | %s | %s | %s | %.2f |" % (
            s, 100 * rows[s]["ta_busy"], rows[s]["l1_accesses_per_clk_per_cu"], rows[s]["l1_accesses_per_vmem_rd"], 100 * div(a["TCP_PENDING_STALL_CYCLES_sum"], a["TCP_GATE_EN1_sum"]),
            "%.0f %%" % (100 * div(fp, a["SQ_INSTS_VALU"])) if have_flops else nc, "%.2f" % tfl if have_flops else nc, "%.3f" % (tfl / 157.3) if have_flops else nc,
            "%.1f %%" % (100 * rows[s]["icache_hit"]) if have_ic else nc, div(a["SQ_INSTS_SALU"], a["SQ_INSTS_VALU"])))
        if not have_flops: rows[s]["fp32_tflops"] = None
        if not have_ic: rows[s]["icache_hit"] = None
if any("TCC_EA0_RDREQ_sum" in acc[s] for s in acc):
    lines += ["", "The L2's memory-side request mix (pass `tccmix`): requests that leave L2 towards Infinity Cache / HBM.  A 32-byte read request moves a quarter of a 128-byte line, a write request that is not `_64B` is a partial (32-byte) write: a stage whose records are scattered 16-byte accesses shows up here as small requests, i.e. it spends memory-side request slots, not bytes.", "",
              "| stage | read requests | of them 32 B | write requests | of them 64 B | mean bytes per request (32 B / 64 B reads, 32 B / 64 B writes) | requests per ns |", "|---|---|---|---|---|---|---|"]
    for s, (calls, tot) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        a = acc[s]
        if "TCC_EA0_RDREQ_sum" not in a: continue
        div = lambda x, y: x / y if y else float("nan")
        rd, rd32, wr, wr64 = a["TCC_EA0_RDREQ_sum"], a["TCC_EA0_RDREQ_32B_sum"], a["TCC_EA0_WRREQ_sum"], a["TCC_EA0_WRREQ_64B_sum"]
        by = rd32 * 32 + (rd - rd32) * 64 + wr64 * 64 + (wr - wr64) * 32
        rows[s]["ea_requests"] = {"read": rd, "read_32B": rd32, "write": wr, "write_64B": wr64, "mean_bytes": div(by, rd + wr), "per_ns": div(rd + wr, tot)}
        lines.append("| %s | %.3g | %.0f %% | %.3g | %.0f %% | %.1f | %.2f |" % (s, rd, 100 * div(rd32, rd), wr, 100 * div(wr64, wr), div(by, rd + wr), div(rd + wr, tot)))
open(os.path.join(ROOT, "profiles", tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
json.dump(rows, open(os.path.join(ROOT, "profiles", tag + "_counters.json"), "w"), indent=1)
print("\n".join(lines))
