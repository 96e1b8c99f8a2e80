"""gpurun_out/shade_sources/ (profiles/collect_shade_sources_r05.sh) -> profiles/r05_shade_sources.md: HBM-side bytes k_shade_stream fetches per path it shades, by what a hit gathers."""
import csv, glob, json, os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "shade_sources")
rows = {}
for name in ("base", "no_sky_nee", "no_light_nee", "no_ec_taps", "tex1x1", "env64"):
    f = glob.glob(os.path.join(G, name, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    fetch = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(f[0])) if "k_shade_stream" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE")
    log = open(os.path.join(G, name + ".log")).read()
    m = re.search(r"closest_rays (\d+)", log)
    paths = int(m.group(1)) * 3 // 2 if m else 0   # the statistics cover the two measured batches, the counters all three (one warm-up batch of the same size)
    ms = re.search(r"'shade': ([0-9.]+)", log); rate = re.search(r"Msamples/s ([0-9.]+)", log); sh = re.search(r"shadow_rays (\d+)", log)
    rows[name] = {"fetched_bytes": 2 * fetch * 1024, "paths_shaded": paths, "bytes_per_path": 2 * fetch * 1024 / max(paths, 1), "shade_ms_two_batches": float(ms.group(1)) if ms else None,
                  "msamples_per_s_under_rocprof": float(rate.group(1)) if rate else None, "shadow_rays_two_batches": int(sh.group(1)) if sh else None}
base = rows.get("base", {}).get("bytes_per_path", float("nan"))
what = {"base": "config 3 as it is", "no_sky_nee": "VPT_FLAG_SKY_MIS clear: no alias entry, no environment texel of the NEE sample (miss lookups remain)", "no_light_nee": "VPT_FLAG_MESH_MIS clear: no light record, no light triangle",
        "no_ec_taps": "VPT_FLAG_ENERGY_COMPENSATION clear: no LUT taps", "tex1x1": "every value texture replaced by its 1x1 mean: no texel lines", "env64": "64 x 32 environment: alias table and texels fit L2"}
lines = ["# k_shade_stream on the atrium: HBM-side bytes fetched per path shaded, by source (r05)", "",
         "`profiles/collect_shade_sources_r05.sh`: 1920x1080, depth 8, 64-frame batches, every sample resident; FETCH_SIZE x 2 (KiB units, gfx950 doubling) summed over the kernel's launches,",
         "divided by the paths the stage shaded (closest-hit rays of the run).  A variant changes what a hit FETCHES; path counts change by a few per cent at most (NEE off shortens no path).", "",
         "| variant | what a hit no longer fetches | fetched B per path | difference to base | shade stage, ms per two 64-frame batches (HIP events) | against base |", "|---|---|---|---|---|---|"]
bms = rows.get("base", {}).get("shade_ms_two_batches") or float("nan")
for k, v in rows.items():
    lines.append("| %s | %s | %.0f | %+.0f | %.2f | %+.1f %% |" % (k, what[k], v["bytes_per_path"], v["bytes_per_path"] - base, v["shade_ms_two_batches"] or float("nan"), 100.0 * ((v["shade_ms_two_batches"] or float("nan")) / bms - 1.0)))
lines += ["", "Records the stage must read per path (ray / throughput / pathLight records, hit record, queue word): 72 B; everything above that is scene gathers that missed L2.", "",
          "Reading: bytes and time do not move together.  Shrinking the environment to fit L2 removes 39 % of the stage's fetched bytes and 0.9 % of its time; switching the sky sample off",
          "altogether (its three draws, alias and texel lookups, six trigonometric evaluations and one BSDF evaluation) removes 35 % of the bytes and 19 % of the time; 1x1 textures 29 % of the",
          "bytes and 10 % of the time (the bilinear decode and the gamma `pow` go with them).  The stage is bound by VALU issue (82 % busy at 77 % lane use), not by what it fetches: the",
          "round-4 verdict's target of <= 2.5 x the record bytes would not have shortened it."]
open(os.path.join(ROOT, "profiles", "r05_shade_sources.md"), "w").write("\n".join(lines) + "\n")
json.dump(rows, open(os.path.join(ROOT, "profiles", "r05_shade_sources.json"), "w"), indent=1)
print("\n".join(lines))
