"""Probe (not a pytest): atrium (Sponza-class) parity at low res, then 1080p timing of both pipelines."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
vpt = importlib.import_module("vulkan-path-tracer_amd")
from oracle import oracle_py as O

def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "atrium"
    t = time.time()
    sc = vpt.scenes.atrium() if which == "atrium" else vpt.scenes.glass_bust()
    depth = 8 if which == "atrium" else 32
    print(which, "tris", sc.triangle_count(), "gen s", round(time.time() - t, 2))
    P = vpt.default_params(max_depth=depth, max_samples=1 << 30)
    W, H = 320, 180
    o = O.Oracle(sc, W, H); o.set_params(P); t = time.time(); o.render(2); print("oracle s", round(time.time() - t, 2))
    ref = o.radiance(); o.close()
    for pipe in (2, 1):
        g = vpt.PathTracer(W, H, pipeline=pipe); t = time.time(); g.set_scene(sc); print("set_scene s", round(time.time() - t, 2)); g.set_params(P); g.render(2)
        img = g.radiance(); st = g.stats(); g.close()
        print("pipeline", pipe, "exact", np.array_equal(img, ref), "differing px", int((np.abs(img - ref).max(axis=2) > 0).sum()), "bvh nodes", st["bvh_nodes"])
    for pipe in (2, 1):
        c = vpt.PathTracer(1920, 1080, pipeline=pipe, count_traversal=True, frames_in_flight=2); c.set_scene(sc); c.set_params(P); c.render(2); cs = c.stats(); c.close()
        g = vpt.PathTracer(1920, 1080, pipeline=pipe, profile=True, frames_in_flight=8); g.set_scene(sc); g.set_params(P)
        g.render(8); g.reset_stats(); t = time.time(); g.render(16); dt = time.time() - t
        st = g.stats(); g.close()
        n = st["samples"]
        print("pipeline", pipe, "Msamples/s", round(n / dt / 1e6, 1), "Mrays/s", round((st["closest_rays"] + st["shadow_rays"]) / dt / 1e6, 1),
              "rays/sample", round(st["closest_rays"] / n, 2), round(st["shadow_rays"] / n, 2),
              "visits closest", round(cs["nodes_visited"] / cs["closest_rays"], 1), round(cs["tris_tested"] / cs["closest_rays"], 2),
              "shadow", round(cs["shadow_nodes_visited"] / max(cs["shadow_rays"], 1), 1), round(cs["shadow_tris_tested"] / max(cs["shadow_rays"], 1), 2))
        print("   ms", {k: round(v, 2) for k, v in st["kernel_ms"].items() if v > 0}, "launches", {k: v for k, v in st["kernel_launches"].items() if v})

    g = vpt.PathTracer(1920, 1080, pipeline=0, frames_in_flight=8); g.set_scene(sc); g.set_params(P)
    g.render(32); g.reset_stats(); t = time.time(); g.render(16); dt = time.time() - t; st = g.stats(); g.close()
    print("pipeline AUTO Msamples/s", round(st["samples"] / dt / 1e6, 1), "launches", {k: v for k, v in st["kernel_launches"].items() if v})

if __name__ == "__main__":
    main()
