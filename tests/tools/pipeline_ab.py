import importlib, os, sys, time, json
sys.path.insert(0,'.')
os.environ.setdefault("VPT_LAB", "1")   # a laboratory tool: loads libvpt_hip_lab.so (include/vpt_lab.h)
vpt = importlib.import_module("vulkan-path-tracer_amd")
import numpy as np
out = {}
for name, sc, depth in (("atrium", vpt.scenes.atrium(), 8), ("bust", vpt.scenes.glass_bust(), 32)):
    imgs = {}
    for pipe in (3, 2, 4):
        g = vpt.PathTracer(1920, 1080, pipeline=pipe, profile=True, frames_in_flight=int(os.environ.get("AB_FRAMES", "0"))); g.set_scene(sc); g.set_params(vpt.default_params(max_depth=depth, max_samples=1<<30))
        F = g.stats()["frames_in_flight"]
        g.render(F); g.reset_stats(); t=time.time(); g.render(F); g.render(F); dt=time.time()-t
        st = g.stats(); imgs[pipe] = g.radiance(); g.close()
        out["%s_pipe%d" % (name, pipe)] = {"msamples": round(st["samples"]/dt/1e6,1), "ms_per_step": round(dt/2*1e3,2), "kernel_ms_per_step": {k: round(v/2,3) for k,v in st["kernel_ms"].items() if v>0}}
        print(name, pipe, out["%s_pipe%d" % (name, pipe)], flush=True)
    out[name+"_equal"] = bool(np.array_equal(imgs[2], imgs[3]) and np.array_equal(imgs[2], imgs[4]))
    print(name, "equal", out[name+"_equal"], flush=True)
json.dump(out, open("gpurun_out/ab.json","w"), indent=1)
