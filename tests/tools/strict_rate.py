"""Probe (not a pytest): what VPT_FLAG_LOCAL_HITS (the structure-independent hit rule) costs in THROUGHPUT on the three BASELINE scenes at
1920x1080, default batch, same box, alternating with the default mode.  Writes gpurun_out/<dir>/strict_rate.json.
    python tests/tools/strict_rate.py [outdir]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
vpt = importlib.import_module("vulkan-path-tracer_amd")
abi = importlib.import_module("vulkan-path-tracer_amd._abi")
out_dir = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r04")
os.makedirs(out_dir, exist_ok=True)
make = {"cornell": lambda: (vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz")), 8), "atrium": lambda: (vpt.scenes.atrium(), 8), "bust": lambda: (vpt.scenes.glass_bust(), 32)}
rows = []
for name in ("cornell", "atrium", "bust"):
    sc, depth = make[name]()
    for rnd in range(2):
        for strict in (False, True):
            P = vpt.default_params(max_depth=depth, max_samples=1 << 30)
            if strict:
                P.flags |= abi.FLAG_LOCAL_HITS
            g = vpt.PathTracer(1920, 1080); g.set_scene(sc); g.set_params(P)
            F = g.stats()["frames_in_flight"]
            for _ in range(2): g.render(F)
            g.reset_stats(); t = time.perf_counter()
            for _ in range(3): g.render(F)
            dt = time.perf_counter() - t; st = g.stats(); g.close()
            row = {"scene": name, "strict_hits": strict, "round": rnd, "msamples_per_s": round(st["samples"] / dt / 1e6, 1), "frames_per_step": F}
            rows.append(row); print(json.dumps(row), flush=True)
summary = {}
for name in ("cornell", "atrium", "bust"):
    d = max(r["msamples_per_s"] for r in rows if r["scene"] == name and not r["strict_hits"])
    s = max(r["msamples_per_s"] for r in rows if r["scene"] == name and r["strict_hits"])
    summary[name] = {"default": d, "strict": s, "cost_pct": round((1 - s / d) * 100, 1)}
print(json.dumps(summary))
json.dump({"rows": rows, "summary": summary}, open(os.path.join(out_dir, "strict_rate.json"), "w"), indent=1)
