"""Probe (not a pytest): the fused kernel's class-specialised instantiation (k_bounce<PLAIN>: no texture code, no environment sampler) against
the general one on the headline workload, same box, same process, alternating: vpt_config.build_flags = VPT_BUILD_GENERAL_KERNELS keeps the
general kernel.  Images must be identical.  Writes gpurun_out/<dir>/plain_ab.json.
    python tests/tools/plain_ab.py [outdir]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
vpt = importlib.import_module("vulkan-path-tracer_amd")
out_dir = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r04")
os.makedirs(out_dir, exist_ok=True)
sc = vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz"))
P = vpt.default_params(max_depth=8, max_samples=0x7fffffff)
rows, imgs = [], {}
for rnd in range(3):
    for name, flags in (("plain", 0), ("general", 2)):
        g = vpt.PathTracer(1920, 1080, build_flags=flags, profile=False)
        g.set_scene(sc); g.set_params(P)
        F = g.stats()["frames_in_flight"]
        for _ in range(2):
            g.render(F)
        g.reset_stats()
        t = time.perf_counter()
        for _ in range(6):
            g.render(F)
        dt = time.perf_counter() - t
        st = g.stats()
        if rnd == 0:
            g.reset(); g.render(4); imgs[name] = g.radiance()
        g.close()
        # per-kernel means from a short profiled pass of the same context configuration
        q = vpt.PathTracer(1920, 1080, build_flags=flags, profile=True)
        q.set_scene(sc); q.set_params(P); q.render(F); q.reset_stats(); q.render(F); q.render(F)
        ps = q.stats(); q.close()
        row = {"variant": name, "round": rnd, "msamples_per_s": round(st["samples"] / dt / 1e6, 1), "frames_per_step": F,
               "primary_ms": round(ps["kernel_ms"]["primary"] / max(ps["kernel_launches"]["primary"], 1), 4),
               "bounce_ms": round(ps["kernel_ms"]["bounce"] / max(ps["kernel_launches"]["bounce"], 1), 4)}
        rows.append(row); print(json.dumps(row), flush=True)
same = bool(np.array_equal(imgs["plain"], imgs["general"]))
print("images identical:", same)
json.dump({"rows": rows, "images_identical": same}, open(os.path.join(out_dir, "plain_ab.json"), "w"), indent=1)
assert same
