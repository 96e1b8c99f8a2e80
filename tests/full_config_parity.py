"""Opt-in (not a pytest: minutes of 256 host cores): a BASELINE config at FULL size, HIP backend vs the CPU oracle, bit for
bit.  config2: Cornell box, 1920x1080, 1024 spp, depth 8.  config3: the 285k-triangle atrium, 1920x1080, 256 spp, depth 8.
Writes gpurun_out/<config>_full_parity.json.
    python tests/full_config_parity.py [config2|config3] [spp]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
vpt = importlib.import_module("vulkan-path-tracer_amd")
from oracle import oracle_py as O

cfg = sys.argv[1] if len(sys.argv) > 1 else "config2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else (1024 if cfg == "config2" else 256)
sc = vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz")) if cfg == "config2" else vpt.scenes.atrium()
P = vpt.default_params(max_depth=8, max_samples=spp)
t = time.time()
g = vpt.PathTracer(1920, 1080); g.set_scene(sc); g.set_params(P); g.render(spp)
img = g.radiance(); st = g.stats(); g.close()
tg = time.time() - t
t = time.time()
o = O.Oracle(sc, 1920, 1080); o.set_params(P); o.render(spp)
ref = o.radiance(); ctr = o.counters(); o.close()
to = time.time() - t
diff = int((np.abs(img - ref).max(axis=2) > 0).sum())
rel = float(np.sqrt(((img[..., :3].astype(np.float64) - ref[..., :3]) ** 2).sum()) / np.sqrt((ref[..., :3].astype(np.float64) ** 2).sum()))
res = {"config": "%s 1920x1080, %d spp, depth 8, base seed 1" % ("cornell" if cfg == "config2" else "atrium (284,880 triangles)", spp), "samples": int(st["samples"]), "bit_exact": bool(np.array_equal(img, ref)),
       "differing_pixels": diff, "rel_l2": rel, "closest_rays_gpu": int(st["closest_rays"]), "closest_rays_oracle": int(ctr["closest"]),
       "gpu_seconds": round(tg, 3), "oracle_seconds": round(to, 1), "oracle_threads": os.cpu_count(), "mean_radiance": float(ref[..., :3].mean())}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "%s_full_parity.json" % cfg), "w"), indent=1)
print(json.dumps(res))
