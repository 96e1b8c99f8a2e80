"""Fixed atrium workload for profiling: 1080p, staged pipeline, 8 frames in flight, 2 batches."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
vpt = importlib.import_module("vulkan-path-tracer_amd")
sc = vpt.scenes.atrium()
g = vpt.PathTracer(1920, 1080, pipeline=2, frames_in_flight=8, profile=True); g.set_scene(sc); g.set_params(vpt.default_params(max_depth=8, max_samples=1 << 30))
g.render(8); g.reset_stats(); t = time.time(); g.render(8); dt = time.time() - t
st = g.stats(); print("Msamples/s", round(st["samples"] / dt / 1e6, 1), {k: round(v, 2) for k, v in st["kernel_ms"].items() if v > 0})
