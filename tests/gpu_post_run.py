"""Fixed workload for profiling the post chain (not a pytest): SIZE=1080p|4k, REPS (default 20) vpt_postprocess calls per
schedule on a synthetic HDR frame (gamma-distributed radiance with 1 % fireflies above the bloom threshold), kernels timed by the
library's HIP events.  Prints one JSON line: per schedule the mean GPU time of the bloom launches and of the tonemap launch per
call, their sum, the launches per call and SURVEY 8d's algorithmic bytes (143 B per pixel of the full frame) over that time."""
import importlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vpt = importlib.import_module("vulkan-path-tracer_amd")
w, h = (3840, 2160) if os.environ.get("SIZE", "1080p") == "4k" else (1920, 1080)
reps = int(os.environ.get("REPS", "20"))
rng = np.random.RandomState(1)
img = np.zeros((h, w, 4), np.float32)
img[..., :3] = rng.gamma(0.4, 4.0, (h, w, 3)); img[rng.rand(h, w) < 0.01, :3] *= 200.0; img[..., 3] = 1.0
g = vpt.PathTracer(w, h, profile=True)
g.set_radiance(img, 1)
out = {"size": [w, h], "reps": reps, "algorithmic_bytes": 143 * w * h}
first = None
for name, schedule in (("reference_passes", 1), ("fused", 0)):
    if os.environ.get("SCHEDULE") not in (None, "", name):
        continue
    pp = vpt.default_post_params(schedule=schedule)
    o8 = g.postprocess(pp)   # warm-up (allocates the mips)
    first = o8 if first is None else first
    assert np.array_equal(o8, first)
    g.reset_stats()
    t = time.time()
    for _ in range(reps):
        g.postprocess(pp)
    wall = (time.time() - t) / reps * 1e3
    st = g.stats()
    bloom, tone = st["kernel_ms"]["bloom"] / reps, st["kernel_ms"]["tonemap"] / reps
    out[name] = {"bloom_ms": round(bloom, 4), "tonemap_ms": round(tone, 4), "gpu_ms": round(bloom + tone, 4),
                 "launches": (st["kernel_launches"]["bloom"] + st["kernel_launches"]["tonemap"]) // reps,
                 "call_wall_ms_incl_readback": round(wall, 3), "algorithmic_GBs": round(143 * w * h / ((bloom + tone) * 1e-3) / 1e9, 1),
                 "frac_of_8TBs": round(143 * w * h / ((bloom + tone) * 1e-3) / 8e12, 3)}
g.close()
print(json.dumps(out))
