"""Probe (not a pytest): VPT_PIPELINE_WHOLE forced on scenes whose BVH lives in memory (k_whole<MEM>, round 6 experiment) — parity against the oracle on the glass-sphere
Cornell box, then one frame per call (blocking and pipelined) and in-batch throughput on the atrium and the bust against AUTO (the streams + finisher).
    python tests/tools/whole_mem_probe.py"""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
vpt = importlib.import_module("vulkan-path-tracer_amd")
from oracle import oracle_py as O
WHOLE = vpt._abi.PIPELINE_WHOLE

sc = vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box_glass.npz"))
for depth, frames in ((6, 3), (40, 2)):
    P = vpt.default_params(max_depth=depth)
    o = O.Oracle(sc, 160, 90); o.set_params(P); o.render(frames); ref = o.radiance(); ctr = o.counters(); o.close()
    g = vpt.PathTracer(160, 90, pipeline=WHOLE, frames_in_flight=frames); g.set_scene(sc); g.set_params(P); g.render(frames)
    img = g.radiance(); st = g.stats(); g.close()
    print(json.dumps({"parity_depth": depth, "bit_exact": bool(np.array_equal(img, ref)), "closest_gpu": int(st["closest_rays"]), "closest_oracle": int(ctr["closest"]), "launches": {k: int(v) for k, v in st["kernel_launches"].items() if v}}), flush=True)

make = {"atrium": lambda: (vpt.scenes.atrium(), 8), "bust": lambda: (vpt.scenes.glass_bust(), 32)}
for name in ("atrium", "bust"):
    scn, depth = make[name]()
    row = {"scene": name}
    for label, pipe in (("auto", 0), ("whole_mem", WHOLE)):
        g = vpt.PathTracer(1920, 1080, frames_in_flight=1, pipeline=pipe)
        g.set_scene(scn); g.set_params(vpt.default_params(max_depth=depth, max_samples=1 << 30))
        for _ in range(4):
            g.render(1); g.postprocess()
        N = 30
        t0 = time.perf_counter()
        for _ in range(N):
            g.render(1); g.postprocess()
        r = {"blocking_frame_ms": round((time.perf_counter() - t0) / N * 1e3, 3)}
        for in_flight in (2, 3):
            for _ in range(8):
                g.render_async(1); g.postprocess_device()
            g.wait()
            t0 = time.perf_counter(); tickets = []
            for _ in range(N * 2):
                g.render_async(1); tickets.append(g.postprocess_device())
                if len(tickets) >= in_flight:
                    g.wait(tickets[-in_flight])
            g.wait()
            r["async_%d_in_flight_ms" % in_flight] = round((time.perf_counter() - t0) / (N * 2) * 1e3, 3)
        g.close()
        # in-batch throughput at 64 frames per batch
        g = vpt.PathTracer(1920, 1080, frames_in_flight=64, pipeline=pipe)
        g.set_scene(scn); g.set_params(vpt.default_params(max_depth=depth, max_samples=1 << 30))
        g.render(64); g.reset_stats(); t0 = time.perf_counter(); g.render(128); dt = time.perf_counter() - t0
        r["msamples_per_s_64_frames"] = round(g.stats()["samples"] / dt / 1e6, 1)
        g.close()
        row[label] = r
    print(json.dumps(row), flush=True)
