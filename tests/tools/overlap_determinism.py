"""Probe (not a pytest): the staged pipeline on two streams is deterministic and equal to the one-kernel-at-a-time schedule — atrium and
glass bust at 1080p, 72 frames in three batches, two overlapped runs and one serial (vpt_config.profile) run compared bit for bit."""
import importlib, os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
vpt = importlib.import_module("vulkan-path-tracer_amd")
for name, sc, depth in (("atrium", vpt.scenes.atrium(), 8), ("bust", vpt.scenes.glass_bust(), 32)):
    imgs = []
    for prof in (False, False, True):   # two overlapped runs and one serial (profile = one kernel at a time)
        g = vpt.PathTracer(1920, 1080, frames_in_flight=24, profile=prof); g.set_scene(sc); g.set_params(vpt.default_params(max_depth=depth, max_samples=1 << 30))
        g.render(72); imgs.append(g.radiance()); g.close()
    print(name, "overlap run 1 == run 2:", np.array_equal(imgs[0], imgs[1]), "| overlap == serial:", np.array_equal(imgs[0], imgs[2]), "| mean", float(imgs[0][..., :3].mean()))
