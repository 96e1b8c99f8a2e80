"""Probe (not a pytest): what the fixed cost of a whole-path launch (kernels_path.hip k_whole) is made of.  Cornell box at 1920x1080; the launch
under HIP events at 1 / 2 / 4 frames per batch for max_depth 1, 2, 4, 8, 16: per depth, the marginal cost per frame and the fixed cost per
launch (a straight line through the three sizes).  If the fixed part is the loop's own depth — the last paths' rounds on emptying waves —
it grows with max_depth while the launch ramp does not.  Writes gpurun_out/<dir>/whole_tail.json.     python tests/tools/whole_tail.py [outdir]"""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
os.environ.setdefault("VPT_LAB", "1")   # a laboratory tool: loads libvpt_hip_lab.so (include/vpt_lab.h)
vpt = importlib.import_module("vulkan-path-tracer_amd")
out_dir = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r04")
os.makedirs(out_dir, exist_ok=True)
sc = vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz"))
rows = []
for depth in (1, 2, 4, 8, 16):
    P = vpt.default_params(max_depth=depth, max_samples=0x7fffffff)
    us, rays = {}, {}
    for F in (1, 2, 4):
        g = vpt.PathTracer(1920, 1080, frames_in_flight=F, profile=True)
        g.set_scene(sc); g.set_params(P)
        for _ in range(3):
            g.render(F)
        g.reset_stats()
        n = 30
        for _ in range(n):
            g.render(F)
        st = g.stats(); g.close()
        us[F] = st["kernel_ms"]["primary"] / n * 1e3
        rays[F] = st["closest_rays"] / st["samples"]
    slope, fixed = np.polyfit([1, 2, 4], [us[1], us[2], us[4]], 1)
    row = {"max_depth": depth, "launch_us": {str(k): round(v, 1) for k, v in us.items()}, "us_per_frame": round(float(slope), 1), "fixed_us_per_launch": round(float(fixed), 1),
           "closest_rays_per_sample": round(rays[4], 3)}
    rows.append(row); print(json.dumps(row), flush=True)
json.dump(rows, open(os.path.join(out_dir, "whole_tail.json"), "w"), indent=1)
