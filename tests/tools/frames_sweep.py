"""Probe (not a pytest): throughput vs RESIDENT frames for the BASELINE scenes whose BVH lives in memory at 1920x1080, batches of F frames
(226 = the default batch; 904 = four times that), vpt_config.resident_frames = K: K = F keeps every sample resident (448M paths at 226, shrinking
launches), smaller K regenerates paths by refill (kernels_stream.hip k_refill_plan: behind every shade stage the room ended paths left in the next
ray queue is filled with a contiguous block of the batch's next samples).  Also the device memory each configuration holds.  Images are compared
(first round, 24-frame batches) against K = batch.  Writes gpurun_out/<dir>/frames_sweep.json.
    python tests/tools/frames_sweep.py [outdir] [scenes]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import ctypes as C
hip = C.CDLL("libamdhip64.so")


def free_bytes():
    f, t = C.c_size_t(0), C.c_size_t(0)
    hip.hipMemGetInfo(C.byref(f), C.byref(t))
    return f.value
vpt = importlib.import_module("vulkan-path-tracer_amd")
out_dir = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r04")
os.makedirs(out_dir, exist_ok=True)
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["atrium", "bust"]
make = {"cornell": lambda: (vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz")), 8), "atrium": lambda: (vpt.scenes.atrium(), 8), "bust": lambda: (vpt.scenes.glass_bust(), 32)}
rows = []
for name in which:
    sc, depth = make[name]()
    P = vpt.default_params(max_depth=depth, max_samples=1 << 30)
    ref = None
    for F, K in ((226, 226), (226, 128), (226, 64), (226, 32), (226, 16), (904, 226), (904, 113), (904, 32)):
        free0 = free_bytes()
        g = vpt.PathTracer(1920, 1080, frames_in_flight=F, resident_frames=K); g.set_scene(sc); g.set_params(P)
        g.render(24)
        img = g.radiance()
        if ref is None:
            ref = img
        same = bool(np.array_equal(img, ref))
        g.reset()
        for _ in range(2 if F == 226 else 1): g.render(F)
        free1 = free_bytes()
        g.reset_stats(); t = time.perf_counter(); n = 4 if F == 226 else 2
        for _ in range(n): g.render(F)
        dt = time.perf_counter() - t; st = g.stats(); g.close()
        launches = sum(st["kernel_launches"][k] for k in ("primary", "bounce", "extend"))
        row = {"scene": name, "batch_frames": F, "resident_frames": st["resident_frames"], "resident_paths_M": round(st["resident_frames"] * 1920 * 1080 / 1e6, 1),
               "msamples_per_s": round(st["samples"] / dt / 1e6, 1), "device_GB": round((free0 - free1) / 2 ** 30, 2), "bounce_launches_per_batch": launches / n,
               "image_identical_to_all_resident": same}
        rows.append(row); print(json.dumps(row), flush=True)
        assert same
json.dump(rows, open(os.path.join(out_dir, "frames_sweep.json"), "w"), indent=1)
