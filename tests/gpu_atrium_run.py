"""Fixed workload for profiling (not a pytest): SIZE (default 1920x1080), depth 8 (atrium) / 32 (bust), FRAMES frames per batch (default 64; 0 = the library's own schedule:
4 x 226 frames per batch with 113 resident at 1080p), RESIDENT = vpt_config.resident_frames, 1 warm-up batch + 2 measured.
    PIPE=2 (staged on streams, default) | 3 (round 1's stage kernels) | 1 (fused) | 0 (the library chooses);  SCENE=atrium|bust|cornell"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vpt = importlib.import_module("vulkan-path-tracer_amd")
pipe = int(os.environ.get("PIPE", "2"))
which = os.environ.get("SCENE", "atrium")
F = int(os.environ.get("FRAMES", "64"))
if which == "cornell":   # the headline scene (LDS-resident: the library picks the fused pipeline whatever PIPE says unless PIPE forces a staged one)
    sc = vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz"))
else:
    variant = os.environ.get("VARIANT", "")   # what the shade stage has to fetch, not how many paths it shades: tex1x1 = every value texture replaced by its 1x1 mean, env64 = a 64x32 environment (fits L2)
    sc = (vpt.scenes.atrium(env_size=(64, 32)) if "env64" in variant else vpt.scenes.atrium()) if which == "atrium" else vpt.scenes.glass_bust()
    if "tex1x1" in variant:
        import numpy as np
        sc.textures = [t if t.shape[0] * t.shape[1] == 1 else np.round(t.reshape(-1, t.shape[2]).mean(0)).astype(np.uint8).reshape(1, 1, -1) for t in sc.textures]
W, H = (int(v) for v in os.environ.get("SIZE", "1920x1080").split("x"))
g = vpt.PathTracer(W, H, pipeline=pipe, frames_in_flight=F, profile=os.environ.get("PROFILE", "1") == "1", resident_frames=int(os.environ.get("RESIDENT", "0"))); g.set_scene(sc)
P = vpt.default_params(max_depth=32 if which == "bust" else 8, max_samples=1 << 30)
P.flags &= ~int(os.environ.get("CLEAR_FLAGS", "0"))   # section costs: 1 sky NEE, 2 light NEE, 16 energy-compensation taps, 8 -> geometry normals
P.flags |= int(os.environ.get("SET_FLAGS", "0"))
g.set_params(P)
if F == 0: F = g.stats()["frames_in_flight"]   # FRAMES=0: the library's own batch (long batches with refill for these scenes)
g.render(F); g.reset_stats(); t = time.time(); g.render(2 * F); dt = time.time() - t
st = g.stats(); print("Msamples/s", round(st["samples"] / dt / 1e6, 1), {k: round(v, 2) for k, v in st["kernel_ms"].items() if v > 0}, "closest_rays", st["closest_rays"], "shadow_rays", st["shadow_rays"])
