"""One rank of tests/test_gpu_comm.py::test_library_gather_at_world_n_through_a_stub_rccl (not a pytest).

    VPT_RCCL_STUB_DIR=<dir> LD_PRELOAD=<librccl_stub.so> python comm_worker.py <rank> <world> <root> <width> <height> <frames> <id file> <out.npy>

Renders its shard of the glass-sphere Cornell scene and calls the library's real vpt_comm_init / vpt_comm_gather_shards; the root then
writes the assembled image and what vpt_comm_get_info reports.  No torch, no other control plane: rank `root` writes the 128-byte
communicator id to <id file>, the others wait for it."""
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
vpt = importlib.import_module("vulkan-path-tracer_amd")
rank, world, root, w, h, frames = (int(v) for v in sys.argv[1:7])
id_file, out = sys.argv[7], sys.argv[8]
sc = vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box_glass.npz"))
g = vpt.PathTracer(w, h, shard_rank=rank, shard_count=world)
g.set_scene(sc); g.set_params(vpt.default_params(max_depth=5))
if rank == root:
    ident = (C.c_ubyte * 128)()
    assert g.lib.vpt_comm_unique_id(ident) == 0
    open(id_file + ".tmp", "wb").write(bytes(ident)); os.rename(id_file + ".tmp", id_file)
for _ in range(30000):
    if os.path.exists(id_file):
        break
    time.sleep(0.001)
raw = open(id_file, "rb").read()
assert g.lib.vpt_comm_init(g.ctx, raw, rank, world) == 0, g.lib.vpt_last_error(g.ctx)
g.render(frames)
assert g.lib.vpt_comm_gather_shards(g.ctx, root) == 0, g.lib.vpt_last_error(g.ctx)
ci = vpt._abi.CommInfo()
assert g.lib.vpt_comm_get_info(g.ctx, C.byref(ci)) == 0
info = {"nranks": ci.nranks, "rank": ci.rank, "device": ci.device, "library_path": ci.library_path.decode(), "rccl_version_runtime": ci.rccl_version_runtime}
if rank == root:
    np.save(out, g.radiance())
    out8 = g.postprocess()
    np.save(out + ".post.npy", out8)
json.dump(info, open(out + ".rank%d.json" % rank, "w"))
assert g.lib.vpt_comm_destroy(g.ctx) == 0
g.close()
