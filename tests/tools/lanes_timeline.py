"""Probe (not a pytest), meant to run under `rocprofv3 --kernel-trace`: 40 asynchronous 1-frame batches + post on the Cornell box at
1920x1080 (three lanes, captured graphs), so that the kernel trace shows which launches of consecutive frames actually overlap.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/<dir>/tl -o tl -- python tests/tools/lanes_timeline.py"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
vpt = importlib.import_module("vulkan-path-tracer_amd")
sc = vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz"))
g = vpt.PathTracer(1920, 1080, frames_in_flight=1)
g.set_scene(sc); g.set_params(vpt.default_params(max_depth=8, max_samples=0x7fffffff))
prev = 0
for _ in range(40):
    g.render_async(1)
    cur = g.postprocess_device()
    if prev:
        g.wait(prev)
    prev = cur
g.wait()
g.close()
