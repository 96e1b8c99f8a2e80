"""Probe (not a pytest): the whole-path launch (VPT_PIPELINE_WHOLE, kernels_path.hip k_whole) against the per-bounce kernels
(VPT_PIPELINE_FUSED) on the Cornell box at 1920x1080, same box, same process, alternating.
  throughput   Msamples/s at 1 / 4 / 16 / 64 / AUTO-cap frames per batch (blocking vpt_render), depth 8; the glass variant at depth 12 (general kernel)
  latency      per-frame wall clock of the blocking pair and of the asynchronous pair with 1 / 2 / 3 frames in flight, lanes 1-3
Images must be identical.  Writes gpurun_out/<dir>/whole_ab.json.     python tests/tools/whole_ab.py [outdir]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
os.environ.setdefault("VPT_LAB", "1")   # a laboratory tool: loads libvpt_hip_lab.so (include/vpt_lab.h)
vpt = importlib.import_module("vulkan-path-tracer_amd")
A = vpt._abi
out_dir = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r04")
os.makedirs(out_dir, exist_ok=True)
res = {"throughput": [], "latency": [], "kernel": []}
W, H = 1920, 1080
SCENES = (("cornell_box", 8), ("cornell_glass", 12))   # the second: the same 12 triangles with glass walls and a sky (general instantiation)
PIPES = (("fused", A.PIPELINE_FUSED), ("whole", A.PIPELINE_WHOLE))
imgs = {}
for name, depth in SCENES:
    sc = vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz"))
    if name == "cornell_glass":
        sc.materials[0].update(transmission=1.0, roughness=0.05, ior=1.5, base_color=(1, 1, 1))
        sc.materials[2].update(transmission=1.0, roughness=0.3, ior=1.33, medium_density=0.6, medium_anisotropy=0.3, medium_color=(0.9, 0.5, 0.4))
        sc.env = vpt.scenes.sun_sky_env(64, 32, seed=9, sun_peak=100.0)
    P = vpt.default_params(max_depth=depth, max_samples=0x7fffffff)
    for F in ((1, 4, 16, 64, 0) if name == "cornell_box" else (1, 0)):
        for pname, pipe in PIPES:
            g = vpt.PathTracer(W, H, pipeline=pipe, frames_in_flight=F)
            g.set_scene(sc); g.set_params(P)
            FF = g.stats()["frames_in_flight"]
            reps = max(3, min(200, 600 // FF))
            for _ in range(2):
                g.render(FF)
            g.reset_stats()
            t = time.perf_counter()
            for _ in range(reps):
                g.render(FF)
            dt = time.perf_counter() - t
            st = g.stats()
            if F == 4 or (name != "cornell_box" and F == 1):
                g.reset(); g.render(FF); imgs[(name, pname)] = g.radiance()
            g.close()
            row = {"scene": name, "depth": depth, "pipeline": pname, "frames_per_batch": FF, "msamples_per_s": round(st["samples"] / dt / 1e6, 1),
                   "ms_per_frame": round(dt / reps / FF * 1e3, 4)}
            res["throughput"].append(row); print(json.dumps(row), flush=True)
    # the kernels themselves (HIP events, profile mode) at the AUTO cap
    for pname, pipe in PIPES:
        q = vpt.PathTracer(W, H, pipeline=pipe, profile=True)
        q.set_scene(sc); q.set_params(P)
        FF = q.stats()["frames_in_flight"]
        q.render(FF); q.reset_stats(); q.render(FF); q.render(FF)
        ps = q.stats(); q.close()
        row = {"scene": name, "pipeline": pname, "frames_per_batch": FF}
        for k in ("primary", "bounce", "resolve"):
            row[k + "_ms_per_batch"] = round(ps["kernel_ms"][k] / 2, 3)
        res["kernel"].append(row); print(json.dumps(row), flush=True)
same = all(np.array_equal(imgs[(n, "fused")], imgs[(n, "whole")]) for n, _ in SCENES)
print("images identical:", same)
res["images_identical"] = bool(same)

# ---- latency of 1-frame batches (the interactive host): AUTO takes the whole-path launch, VPT_LAB_WHOLE_FRAMES = 0 keeps the per-bounce kernels
sc = vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz"))
P = vpt.default_params(max_depth=8, max_samples=0x7fffffff)
g = vpt.PathTracer(W, H, frames_in_flight=1)
g.set_scene(sc); g.set_params(P)
N = 200


def async_loop(in_flight, n=N):
    for _ in range(8):
        g.render_async(1)
    g.wait()
    t = time.perf_counter(); tickets = []
    for _ in range(n):
        g.render_async(1)
        tickets.append(g.postprocess_device())
        if len(tickets) >= in_flight:
            g.wait(tickets[-in_flight])
    g.wait()
    return (time.perf_counter() - t) / n * 1e3


for mode, whole_frames in (("whole", 1), ("per_bounce", 0), ("whole", 1)):
    g.lab_set(A.LAB_WHOLE_FRAMES, whole_frames)
    for _ in range(10):
        g.render(1); g.postprocess()
    t = time.perf_counter()
    for _ in range(N):
        g.render(1); g.postprocess()
    row = {"mode": mode, "blocking_render_post_ms": round((time.perf_counter() - t) / N * 1e3, 4)}
    t = time.perf_counter()
    for _ in range(N):
        g.render(1)
    row["blocking_render_ms"] = round((time.perf_counter() - t) / N * 1e3, 4)
    for lanes in (1, 2, 3):
        g.lab_set(A.LAB_LANES, lanes)
        for infl in (1, 2, 3):
            row["async_lanes%d_in_flight%d_ms" % (lanes, infl)] = round(async_loop(infl), 4)
    g.lab_set(A.LAB_LANES, 3)
    res["latency"].append(row); print(json.dumps(row), flush=True)
g.close()
json.dump(res, open(os.path.join(out_dir, "whole_ab.json"), "w"), indent=1)
assert same
