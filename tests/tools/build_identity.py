"""Probe (not a pytest): two builds of the library render the BASELINE scenes at FULL size and the frame sums are compared bit for bit.  The round's full-size parity runs
against the oracle (profiles/r05_config*_full_parity.json) were made on the build with source id b0a9152ffe1b4658; kernels changed after them (sample_emissive's single
assignment, the uniform wave index of the stream kernels).  This ties the final build to that one at the sizes the oracle runs took: equal images here + equal-to-oracle there.
    python tests/tools/build_identity.py variants/<a>.so variants/<b>.so [outdir]     (each build renders in its own process: the library is swapped on disk)"""
import importlib, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = (("cornell", 1920, 1080, 8, 256), ("atrium", 1920, 1080, 8, 64), ("atrium", 3840, 2160, 8, 16), ("bust", 1920, 1080, 32, 64))   # scene, size, depth, samples per pixel

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np
    vpt = importlib.import_module("vulkan-path-tracer_amd")
    out = sys.argv[2]
    rows = []
    for name, w, h, depth, spp in CASES:
        sc = vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz")) if name == "cornell" else (vpt.scenes.atrium() if name == "atrium" else vpt.scenes.glass_bust())
        g = vpt.PathTracer(w, h); g.set_scene(sc); g.set_params(vpt.default_params(max_depth=depth, max_samples=1 << 30))
        g.render(spp)
        img = g.radiance(); st = g.stats(); g.close()
        np.save(os.path.join(out, "%s_%dx%d.npy" % (name, w, h)), img)
        rows.append({"scene": name, "size": [w, h], "depth": depth, "spp": spp, "samples": st["samples"], "closest_rays": st["closest_rays"], "shadow_rays": st["shadow_rays"]})
    json.dump(rows, open(os.path.join(out, "rows.json"), "w"))
    sys.exit(0)

import numpy as np
a, b = sys.argv[1], sys.argv[2]
outdir = os.path.join(ROOT, "gpurun_out", sys.argv[3] if len(sys.argv) > 3 else "r05_identity")
product = os.path.join(ROOT, "vulkan-path-tracer_amd", "libvpt_hip.so")
keep = product + ".keep"
shutil.copy(product, keep)
res = {}
try:
    for tag, lib in (("a", a), ("b", b)):
        d = os.path.join(outdir, tag); os.makedirs(d, exist_ok=True)
        shutil.copy(os.path.join(ROOT, lib), product)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", d])
        res[tag] = json.load(open(os.path.join(d, "rows.json")))
finally:
    shutil.move(keep, product)
report = {"a": a, "b": b, "cases": []}
ok = True
for ra, rb in zip(res["a"], res["b"]):
    f = "%s_%dx%d.npy" % (ra["scene"], ra["size"][0], ra["size"][1])
    ia, ib = np.load(os.path.join(outdir, "a", f)), np.load(os.path.join(outdir, "b", f))
    same = bool(np.array_equal(ia.view(np.uint32), ib.view(np.uint32)))
    counters = all(ra[k] == rb[k] for k in ("samples", "closest_rays", "shadow_rays"))
    ok = ok and same and counters
    report["cases"].append(dict(ra, differing_pixels=int((ia.view(np.uint32) != ib.view(np.uint32)).any(axis=-1).sum()), bit_identical=same, ray_counts_equal=counters))
    os.remove(os.path.join(outdir, "a", f)); os.remove(os.path.join(outdir, "b", f))
report["all_identical"] = ok
json.dump(report, open(os.path.join(outdir, "build_identity.json"), "w"), indent=1)
print(json.dumps(report))
sys.exit(0 if ok else 1)
