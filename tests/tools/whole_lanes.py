"""Probe (not a pytest): how 1-frame whole-path launches (kernels_path.hip k_whole) share the chip when frames are pipelined: tile schedule
(VPT_LAB_WHOLE_SCHED) x lanes x frames in flight, steady-state wall clock per frame of vpt_render_async(1) + vpt_postprocess_device on the Cornell box at
1920x1080 depth 8; and the launch itself under HIP events at 1 / 2 / 4 frames per batch.  Writes gpurun_out/<dir>/whole_lanes.json.
    python tests/tools/whole_lanes.py [outdir]"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("VPT_LAB", "1")   # a laboratory tool: loads libvpt_hip_lab.so (include/vpt_lab.h)
vpt = importlib.import_module("vulkan-path-tracer_amd")
A = vpt._abi
out_dir = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r04")
os.makedirs(out_dir, exist_ok=True)
sc = vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz"))
P = vpt.default_params(max_depth=8, max_samples=0x7fffffff)
res = {"launch": [], "policies": []}
SCHEDS = (0x04, 0x34, 0x38, 0x14, 0x04, 0x34)   # VPT_LAB_WHOLE_SCHED: tiles per atomic | static-rounds mode << 4
for F in (1, 2, 4, 32, 226):
    g = vpt.PathTracer(1920, 1080, frames_in_flight=F, profile=True)
    g.set_scene(sc); g.set_params(P)
    for sched in SCHEDS:
        g.lab_set(A.LAB_WHOLE_SCHED, sched)
        for _ in range(3):
            g.render(F)
        g.reset_stats()
        n = 30 if F <= 4 else 6 if F <= 32 else 2
        for _ in range(n):
            g.render(F)
        st = g.stats()
        row = {"frames_per_batch": F, "sched": "0x%02x" % sched, "launch_us": round(st["kernel_ms"]["primary"] / n * 1e3, 2), "us_per_frame": round(st["kernel_ms"]["primary"] / n / F * 1e3, 2),
               "resolve_us": round(st["kernel_ms"]["resolve"] / n * 1e3, 2)}
        res["launch"].append(row); print(json.dumps(row), flush=True)
    g.close()
g = vpt.PathTracer(1920, 1080, frames_in_flight=1)
g.set_scene(sc); g.set_params(P)
N = 200


def async_loop(in_flight, n=N):
    for _ in range(8):
        g.render_async(1); g.postprocess_device()
    g.wait()
    t = time.perf_counter(); tickets = []
    for _ in range(n):
        g.render_async(1)
        tickets.append(g.postprocess_device())
        if len(tickets) >= in_flight:
            g.wait(tickets[-in_flight])
    g.wait()
    return (time.perf_counter() - t) / n * 1e3


for sched in (0x04, 0x34, 0x38, 0x04, 0x34):
    g.lab_set(A.LAB_WHOLE_SCHED, sched)
    for lanes in (1, 2, 3):
        g.lab_set(A.LAB_LANES, lanes)
        row = {"sched": "0x%02x" % sched, "lanes": lanes}
        for infl in (1, 2, 3, 4):
            row["frame_ms_%d_in_flight" % infl] = round(async_loop(infl), 4)
        res["policies"].append(row); print(json.dumps(row), flush=True)
g.close()
json.dump(res, open(os.path.join(out_dir, "whole_lanes.json"), "w"), indent=1)
