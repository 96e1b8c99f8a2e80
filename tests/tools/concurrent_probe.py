"""Probe (not a pytest): do two independent renders of the same scene, issued from two host threads on their own streams, finish sooner
together than one after the other?  If they do, co-resident kernels of different stages use resources one stage leaves idle.
    SCENE=atrium FRAMES=64 python tests/tools/concurrent_probe.py"""
import importlib, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
vpt = importlib.import_module("vulkan-path-tracer_amd")
which = os.environ.get("SCENE", "atrium")
F = int(os.environ.get("FRAMES", "64"))
sc = vpt.scenes.atrium() if which == "atrium" else vpt.scenes.glass_bust()
def make():
    g = vpt.PathTracer(1920, 1080, pipeline=2, frames_in_flight=F); g.set_scene(sc)
    g.set_params(vpt.default_params(max_depth=8 if which == "atrium" else 32, max_samples=1 << 30)); g.render(F); return g
a, b = make(), make()
def run(g, n): g.render(n)
t = time.time(); run(a, 2 * F); run(b, 2 * F); serial = time.time() - t
t = time.time()
ta = threading.Thread(target=run, args=(a, 2 * F)); tb = threading.Thread(target=run, args=(b, 2 * F)); ta.start(); tb.start(); ta.join(); tb.join()
both = time.time() - t
px = 1920 * 1080 * 4 * F / 1e6
print("serial %.3f s = %.1f Msamples/s; concurrent %.3f s = %.1f Msamples/s (%+.1f %%)" % (serial, px / serial, both, px / both, 100 * (serial / both - 1)))
