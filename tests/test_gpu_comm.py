"""The one collective of the path through the C-ABI (include/vpt.h vpt_comm_* / vpt_multi_gather_shards), the bench's N > 1
path on one device, and the error paths that must leave a context usable.  RCCL refuses two ranks on one device, so on
a one-GPU box the communicator is exercised with world 1 (ncclCommInitRank + ncclGather do run) and the N-shard logic
through the single-process peer-copy form; the torchrun launch of bench.py runs the same Python as an 8-GPU node with the
shards staged through host memory (VPT_BENCH_DEVICE hook)."""
import ctypes as C
import importlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def render_whole(vpt, sc, w, h, P, frames):
    g = vpt.PathTracer(w, h); g.set_scene(sc); g.set_params(P); g.render(frames)
    img = g.radiance(); g.close()
    return img


def test_rccl_communicator_world1(vpt, scenes):
    sc = scenes("cornell_box")
    P = vpt.default_params(max_depth=4)
    ref = render_whole(vpt, sc, 96, 54, P, 3)
    g = vpt.PathTracer(96, 54); g.set_scene(sc); g.set_params(P)
    ident = (C.c_ubyte * 128)()
    assert g.lib.vpt_comm_unique_id(ident) == 0 and any(ident)
    assert g.lib.vpt_comm_gather_shards(g.ctx, 0) == -1                      # before init
    assert g.lib.vpt_comm_init(g.ctx, bytes(ident), 1, 2) == -1              # rank / world must match the context's shard
    assert g.lib.vpt_comm_init(g.ctx, bytes(ident), 0, 1) == 0, g.lib.vpt_last_error(g.ctx)
    g.render(3)
    assert g.lib.vpt_comm_gather_shards(g.ctx, 0) == 0, g.lib.vpt_last_error(g.ctx)
    assert np.array_equal(g.radiance(), ref)
    assert g.lib.vpt_comm_destroy(g.ctx) == 0
    g.close()


# world 8: the node's shape — 1080 rows (135 per rank) and 2160 rows (270 per rank) of a narrow image, and a ragged height (1083 = 8 x 135 + 3: ranks 0-2 own 136 rows)
@pytest.mark.parametrize("world,root,h", [(2, 0, 54), (3, 2, 55), (8, 0, 1080), (8, 5, 2160), (8, 0, 1083)])
def test_library_gather_at_world_n_through_a_stub_rccl(vpt, scenes, tmp_path, world, root, h):
    """The C++ gather path itself — vpt_comm_init, vpt_comm_gather_shards (padding, root-only receive buffer, offset r * count per
    rank), the row re-interleave on the root, post-processing of the assembled image — at world > 1: `world` PROCESSES on this box's
    one GPU, each calling the library's real entry points, with tests/tools/rccl_stub.cpp in front of librccl (LD_PRELOAD; RCCL
    itself refuses two ranks on one device).  The stub moves the bytes (through files); everything else is the product's code,
    which until this test had only ever executed with world = 1.  Bit-identical to the unsharded render; RCCL-side facts
    (nranks, rank) come back through vpt_comm_get_info from what the communicator reports."""
    stub = str(tmp_path / "librccl_stub.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "-O2", "-o", stub, os.path.join(ROOT, "tests", "tools", "rccl_stub.cpp")])
    sc = scenes("cornell_box_glass")
    P = vpt.default_params(max_depth=5)
    w, frames = (80, 2) if h < 1000 else (16, 1)
    ref = render_whole(vpt, sc, w, h, P, frames)
    g = vpt.PathTracer(w, h); g.set_scene(sc); g.set_params(P); g.render(frames); ref8 = g.postprocess(); g.close()
    env = dict(os.environ, LD_PRELOAD=stub, VPT_RCCL_STUB_DIR=str(tmp_path))
    out = str(tmp_path / "img.npy")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "tools", "comm_worker.py"), str(r), str(world), str(root), str(w), str(h), str(frames),
                               str(tmp_path / "id.bin"), out], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    assert np.array_equal(np.load(out), ref) and np.array_equal(np.load(out + ".post.npy"), ref8)
    for r in range(world):
        info = json.load(open(out + ".rank%d.json" % r))
        assert info["nranks"] == world and info["rank"] == r and info["library_path"].endswith("librccl_stub.so")


def test_comm_info_names_the_mapped_rccl(vpt, scenes):
    """vpt_comm_get_info: which librccl the process mapped (PyTorch's own wins over /opt/rocm's when torch was imported first), its
    version next to the one this library was compiled against, and the communicator as RCCL sees it."""
    g = vpt.PathTracer(32, 18); g.set_scene(scenes("cornell_box"))
    ci = vpt._abi.CommInfo()
    assert g.lib.vpt_comm_get_info(g.ctx, C.byref(ci)) == 0
    assert ci.nranks == 0 and ci.rccl_version_runtime // 10000 == ci.rccl_version_compiled // 10000 and b"rccl" in ci.library_path
    ident = (C.c_ubyte * 128)()
    assert g.lib.vpt_comm_unique_id(ident) == 0 and g.lib.vpt_comm_init(g.ctx, bytes(ident), 0, 1) == 0
    assert g.lib.vpt_comm_get_info(g.ctx, C.byref(ci)) == 0 and (ci.nranks, ci.rank) == (1, 0)
    bus = C.create_string_buffer(64)
    assert g.lib.vpt_device_identity(g.ctx, bus, 64) == 0 and bus.value.count(b":") >= 2
    assert g.lib.vpt_device_identity(g.ctx, bus, 8) == -1
    g.lib.vpt_comm_destroy(g.ctx); g.close()


@pytest.mark.parametrize("n,h", [(2, 54), (3, 55), (8, 61)])
def test_multi_gather_shards_equals_one_context(vpt, scenes, n, h):
    sc = scenes("cornell_box_glass")
    P = vpt.default_params(max_depth=5)
    ref = render_whole(vpt, sc, 80, h, P, 2)
    parts = []
    for r in range(n):
        s = vpt.PathTracer(80, h, shard_rank=r, shard_count=n); s.set_scene(sc); s.set_params(P); s.render(2)
        parts.append(s)
    arr = (C.c_void_p * n)(*[p.ctx for p in parts])
    root = n - 1                                                            # any shard can be the root
    assert parts[0].lib.vpt_multi_gather_shards(arr, n, root) == 0, parts[root].lib.vpt_last_error(parts[root].ctx)
    assert np.array_equal(parts[root].radiance(), ref)
    out8 = parts[root].postprocess()
    g = vpt.PathTracer(80, h); g.set_scene(sc); g.set_params(P); g.render(2)
    assert np.array_equal(out8, g.postprocess()); g.close()
    assert parts[0].lib.vpt_multi_gather_shards(arr, n - 1, 0) == -1         # count must equal shard_count
    for p in parts:
        p.close()


def test_failed_resize_and_bad_material_leave_the_context_usable(vpt, scenes):
    abi = importlib.import_module("vulkan-path-tracer_amd._abi")
    sc = scenes("cornell_box")
    P = vpt.default_params(max_depth=3)
    ref = render_whole(vpt, sc, 64, 36, P, 2)
    g = vpt.PathTracer(64, 36); g.set_scene(sc); g.set_params(P)
    assert g.lib.vpt_resize(g.ctx, 70000, 70000) == -1                       # rejected before anything is freed
    g.render(2)
    assert np.array_equal(g.radiance(), ref)
    m = g.get_material(0)
    bad = abi.Material.from_buffer_copy(bytes(m)); bad.base_color_texture = 9999
    assert g.lib.vpt_set_material(g.ctx, 0, C.byref(bad)) == -1              # texture index out of range: nothing changes
    assert bytes(g.get_material(0)) == bytes(m)
    g.reset(); g.render(2)
    assert np.array_equal(g.radiance(), ref)
    g.close()


def test_rejected_scene_keeps_the_current_one(vpt, scenes):
    sc = scenes("cornell_box")
    P = vpt.default_params(max_depth=3)
    ref = render_whole(vpt, sc, 64, 36, P, 2)
    g = vpt.PathTracer(64, 36); g.set_scene(sc); g.set_params(P)
    desc, keep = sc.to_desc()
    desc.materials[0].base_color_texture = 12345                            # invalid description
    assert g.lib.vpt_set_scene(g.ctx, C.byref(desc)) == -1
    del keep
    g.render(2)                                                             # the old scene is still there
    assert np.array_equal(g.radiance(), ref)
    g.close()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def test_bench_two_ranks_on_one_device():
    env = dict(os.environ, VPT_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames-in-flight", "2"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak" and line["config"]["workload"] == "cornell_1080p_d8"
    assert line["roofline"]["frac"] <= 1.0 and line["roofline"]["bound"] in ("valu", "hbm")


def test_bench_gpus_8_dry_run_on_one_device():
    """`bench.py --gpus 8` as the driver launches it on a node, on this box's one device through the one-device hook (eight processes share the GPU, the gather is host-staged):
    the emitted line must say 8 ranks, the communicator must report 8, and weak mode must keep the per-rank frame count of the N = 1 run."""
    env = dict(os.environ, VPT_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--frames-in-flight", "2", "--no-cpu-baseline", "--no-extra-workloads", "--no-latency", "--no-live-pmc"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["value"] > 0
    assert line["rccl"]["nranks"] == 8
    assert line["config"]["frames_per_step_per_gpu"] == 2 and line["config"]["workload"] == "cornell_1080p_d8"
    # whole-job samples of the timed region: 8 ranks x 2 frames x 135 rows each = 2 whole 1080p frames per step... weak mode renders frames_per_step_per_gpu frames of the rank's rows
    assert abs(line["value"] * line["ms_per_step"] * 1e-3 * 1e6 - 2 * 1920 * 1080) < 1e-3 * 2 * 1920 * 1080 * 8


def test_bench_strong_scaling_splits_a_fixed_job():
    """bench.py --scaling strong: the job is steps x 129 whole frames whatever N; two ranks (on this box's one device, host-staged
    gather) trace together exactly the samples one rank does alone, and the line says so."""
    totals = []
    for n in (1, 2):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        args = [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "1", "--frames-in-flight", "70", "--scaling", "strong",
                "--no-cpu-baseline", "--no-extra-workloads", "--no-latency"]
        if n == 1:
            cmd = [sys.executable] + args
        else:
            env["VPT_BENCH_DEVICE"] = "0"
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args
        p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
        assert p.returncode == 0, p.stderr[-3000:]
        line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        assert line["scaling"] == "strong" and line["n_gpus"] == n and line["config"]["batches_per_gpu"] == [70, 59]   # 129 frames in batches of <= 70
        totals.append(round(line["value"] * line["ms_per_step"] * 1e-3 * line["steps"] * 1e6))
    assert abs(totals[0] - 129 * 1920 * 1080) < 1e-3 * totals[0] and abs(totals[1] - totals[0]) < 1e-3 * totals[0]


def test_bench_line_contract_single_gpu():
    """`python bench.py` at N = 1 (short: 2 steps of 4 frames, CPU sample of ~1 s, no extra workloads): ONE JSON line with the driver's
    fields, a roofline object whose fraction is a fraction, and the CPU baseline object."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--frames-in-flight", "4", "--no-extra-workloads", "--cpu-seconds", "1"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["warmup"] == 1 and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["unit"] == "Msamples/s" and line["dtype"] == "f32" and line["data"] == "synthetic" and line["config"]["workload"] == "cornell_1080p_d8"
    assert abs(line["value"] - 2 * 4 * 1920 * 1080 / (2 * line["ms_per_step"] * 1e-3) / 1e6) < 0.01 * line["value"]   # value = samples / timed seconds
    r = line["roofline"]
    assert r["bound"] in ("valu", "hbm") and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["kernel"] in r["kernels"] and r["traversal"]["nodes_per_closest_ray"] > 0
    lat = line["latency"]   # the reference's per-frame call pattern, blocking (vpt_render(ctx, 1) + vpt_postprocess) and asynchronous
    assert lat["blocking_frame_ms"] > 0 and abs(lat["blocking_frame_ms"] - lat["render_1spp_ms"] - lat["postprocess_ms"]) < 1e-3
    assert 0 < lat["frame_ms"] <= lat["blocking_frame_ms"] * 1.05 and lat["graph"] is True
    assert line["set_scene"]["set_scene_ms"] > 0
    # the counters are this run's own (bench.py live_pmc: rocprofv3 --pmc child passes) or, where rocprofv3 is unavailable, the committed file's under its source-id rule
    if r["pmc"]["source"].startswith("live"):
        assert r["pmc"]["corrections"] and r["traffic"] > 0 and 0 < r["valu"]["frac"] <= 1.0 and "committed_file" in r["pmc"]
    else:
        assert r["pmc"]["rule"] and (r["pmc"]["stale"] or r["valu"] is None or 0 < r["valu"]["frac"] <= 1.0)
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Msamples/s" and c["value"] > 0 and c["cores"] >= 1 and "frames" in c["sample"]
