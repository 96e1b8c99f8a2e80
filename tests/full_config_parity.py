"""Opt-in (not a pytest: minutes of 256 host cores): a BASELINE config at its OWN size, HIP backend vs the CPU oracle, bit for bit.
    config2         Cornell box, 1920x1080, depth 8                      (default 1024 spp)
    config3         atrium (253,002 triangles), 1920x1080, depth 8        (default 256 spp)
    config3_strict  the same with VPT_FLAG_LOCAL_HITS on both sides: the structure-independent hit rule, expected 0 differing pixels
    config4         atrium, 3840x2160, depth 8: one rank's worth of BASELINE config 4, whole frames   (default 1 spp)
    config4_strict  the same with VPT_FLAG_LOCAL_HITS
    config5         glass bust, 1920x1080, depth 32, + bloom / tonemap post, RGBA8 compared too     (default 8 spp)
Writes gpurun_out/<config>_full_parity.json.
    python tests/full_config_parity.py <config> [spp]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
vpt = importlib.import_module("vulkan-path-tracer_amd")
abi = importlib.import_module("vulkan-path-tracer_amd._abi")
from oracle import oracle_py as O

cfg = sys.argv[1] if len(sys.argv) > 1 else "config2"
DEFAULT_SPP = {"config2": 1024, "config3": 256, "config3_strict": 256, "config4": 1, "config4_strict": 1, "config5": 8}
spp = int(sys.argv[2]) if len(sys.argv) > 2 else DEFAULT_SPP[cfg]
W, H = (3840, 2160) if cfg.startswith("config4") else (1920, 1080)
depth = 32 if cfg == "config5" else 8
if cfg == "config2":
    sc, what = vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz")), "cornell"
elif cfg == "config5":
    sc, what = vpt.scenes.glass_bust(), "glass bust (%d triangles)" % vpt.scenes.glass_bust().triangle_count()
else:
    sc = vpt.scenes.atrium(); what = "atrium (%d triangles)" % sc.triangle_count()
P = vpt.default_params(max_depth=depth, max_samples=spp)
if cfg.endswith("_strict"):
    P.flags |= abi.FLAG_LOCAL_HITS
t = time.time()
g = vpt.PathTracer(W, H); t_create = time.time() - t
g.set_scene(sc); g.set_params(P); t_scene = time.time() - t - t_create
g.render(spp); t_render = time.time() - t - t_create - t_scene   # incl. growing the path buffers to the batch size (hipMalloc of tens of GB) on the first batch
img = g.radiance(); st = g.stats()
out8 = g.postprocess() if cfg == "config5" else None
g.close()
tg = time.time() - t
t = time.time()
o = O.Oracle(sc, W, H); o.set_params(P); o.render(spp)
ref = o.radiance(); ctr = o.counters(); o.close()
to = time.time() - t
diff = int((np.abs(img - ref).max(axis=2) > 0).sum())
rel = float(np.sqrt(((img[..., :3].astype(np.float64) - ref[..., :3]) ** 2).sum()) / np.sqrt((ref[..., :3].astype(np.float64) ** 2).sum()))
res = {"config": "%s %dx%d, %d spp, depth %d, base seed 1%s" % (what, W, H, spp, depth, ", VPT_FLAG_LOCAL_HITS" if cfg.endswith("_strict") else ""),
       "samples": int(st["samples"]), "bit_exact": bool(np.array_equal(img, ref)),
       "differing_pixels": diff, "rel_l2": rel, "tolerance_rel_l2": 1e-4, "closest_rays_gpu": int(st["closest_rays"]), "closest_rays_oracle": int(ctr["closest"]),
       "pipeline_kernels": {k: int(v) for k, v in st["kernel_launches"].items() if v},
       "gpu_seconds": round(tg, 3), "gpu_seconds_split": {"create": round(t_create, 3), "set_scene_and_params": round(t_scene, 3), "render_incl_buffer_growth": round(t_render, 3), "read_back_stats_post_close": round(tg - t_create - t_scene - t_render, 3)},
       "source_id": importlib.import_module("vulkan-path-tracer_amd._build").source_id(), "finish_paths": int(st.get("finish_paths", 0)),
       "batch_frames": int(st["frames_in_flight"]), "resident_frames": int(st["resident_frames"]), "oracle_seconds": round(to, 1), "oracle_threads": os.cpu_count(), "mean_radiance": float(ref[..., :3].mean())}
if out8 is not None:
    ref8, _ = O.postprocess(ref, vpt.default_post_params())
    res["post_rgba8_differing_bytes"] = int((out8 != ref8).sum())
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "%s_full_parity.json" % cfg), "w"), indent=1)
print(json.dumps(res))
