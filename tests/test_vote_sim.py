"""tests/tools/vote_sim (the host model of one wave of the vote-scheduled traversal kernels, profiles/NOTEBOOK_r1_r5.md section 4): every scheduling policy it prices must
return the hits of the product's policy — it walks the PRODUCT tree (csrc/bvh_build.cpp) with the device's quantised slab test restated — and the product
policy's lane use must stay where the counters put the kernels (roughly half of the 64 lanes), or the tool no longer gauges anything.  CPU only."""
import os
import re
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_policy_returns_the_product_policys_hits(vpt, oracle, scenes, tmp_path):
    exe = str(tmp_path / "vote_sim")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-march=x86-64-v3", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "tools", "vote_sim.cpp"),
                           os.path.join(ROOT, "vulkan-path-tracer_amd", "csrc", "bvh_build.cpp"), "-o", exe, "-lpthread"])
    sc = scenes("viking_room")
    o = oracle.Oracle(sc, 8, 8); tris = o.triangles(); o.close()
    tris.tofile(str(tmp_path / "tris.bin"))
    rng = np.random.default_rng(9)
    n = 6000
    idx = rng.integers(0, len(tris), n)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.zeros((n, 10), np.float32)
    rays[:, 0:3] = tris[idx, 0:3] + tris[idx, 3:6] * 0.3 + tris[idx, 6:9] * 0.3 + d * 1e-3
    rays[:, 3] = 1e-4; rays[:, 4:7] = d; rays[:, 7] = 1e6; rays[:, 8] = 1e6
    rays[:, 9] = np.full(n, 0xffffffff, np.uint32).view(np.float32)
    rays.tofile(str(tmp_path / "rays.bin"))
    for kind in ("closest", "any"):
        p = subprocess.run([exe, str(tmp_path / "tris.bin"), str(tmp_path / "rays.bin"), kind], capture_output=True, text=True)
        assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
        rows = [l for l in p.stdout.splitlines() if "mismatches" in l]
        assert len(rows) >= 12
        assert all(l.rstrip().endswith("mismatches 0") for l in rows), [l for l in rows if not l.rstrip().endswith("mismatches 0")]
        use = float(re.search(r"lane use\s+([0-9.]+) %", rows[0]).group(1))     # the product policy
        assert 35.0 < use < 80.0, rows[0]
        pool = [l for l in rows if l.startswith("pool 128, fetch at 32")][0]
        assert float(re.search(r"lane use\s+([0-9.]+) %", pool).group(1)) > use + 15.0   # the model's one large effect stays visible
