"""Probe (not a pytest): throughput vs frames in flight (resident paths) for the three BASELINE scenes."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
vpt = importlib.import_module("vulkan-path-tracer_amd")
scenes = {"cornell": (vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz")), 8), "atrium": (vpt.scenes.atrium(), 8), "bust": (vpt.scenes.glass_bust(), 32)}
out = {}
for name, (sc, depth) in scenes.items():
    for F in (64, 128):
        g = vpt.PathTracer(1920, 1080, frames_in_flight=F); g.set_scene(sc); g.set_params(vpt.default_params(max_depth=depth, max_samples=1 << 30))
        for _ in range(5): g.render(F)
        g.reset_stats(); t = time.time(); n = max(2, 128 // F)
        for _ in range(n): g.render(F)
        dt = time.time() - t; st = g.stats(); g.close()
        out["%s_F%d" % (name, F)] = round(st["samples"] / dt / 1e6, 1)
        print(name, F, out["%s_F%d" % (name, F)], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "frames_sweep.json"), "w"), indent=1)
