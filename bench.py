#!/usr/bin/env python
"""bench.py — Msamples/s of the wavefront path tracer on BASELINE.json's metric config.

Workload (config.workload = "cornell_1080p_d8"): BASELINE.json configs[1] — Assets/CornellBox as shipped
(diffuse walls + emissive quad, strength 50), black environment, 1920x1080, max depth 8, 1 sample per
pixel per frame, base seed 1.  A *step* is one wavefront batch of the hot path:
every rank renders `frames_per_step` consecutive frames of its own rows (fused primary bounce, then one
fused bounce kernel — or extend/shade/connect for scenes too big for LDS — per bounce until the ray queue is
empty, then resolve).  Rows are dealt round-robin over ranks, each rank keeps ~32M paths resident, so
per-GPU work per step is fixed (weak scaling) and

    value = (samples all ranks traced in the K timed steps) / (max over ranks of the wall time)

with scene, BVH, path state and accumulation image resident in HBM before the timed region.  The timed
region ends with the single collective of the path: one gather of the finished row shards (RCCL, backend
"nccl") and the row re-interleave on rank 0.

Extra JSON objects (see DESIGN.md §6 for the byte accounting):
  roofline     — for the kernel with the largest share of GPU time in the timed region: algorithmic HBM
                 bytes per launch / mean launch duration (HIP events recorded by the library on the stream
                 it launches on), against the 8 TB/s HBM3E peak.  `kernels` lists every stage.
  cpu_baseline — the CPU oracle (oracle/, a port of the reference shaders; the reference has no CPU path)
                 on this box's host cores, on a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured copy
WIDTH, HEIGHT, MAX_DEPTH, BASE_SEED = 1920, 1080, 8, 1

# Algorithmic bytes per unit for each stage: state words actually read/written per path or
# ray by the algorithm with this build's struct sizes; BVH node/triangle visits are measured, not assumed.
TRI_BYTES = 48   # triangle record; the node size (64 B quantised, 128 B fp32 when the BVH rides in LDS) comes from vpt_stats
EXTEND_FIXED = 4 + 24 + 20          # queue id, origin+direction in, hit record out
SHADE_IN = 4 + 16 + 16 + 16 + 20    # queue id, records A (origin|rng), B (dir|depth), T (throughput|pdf), hit record
SHADE_ALIVE_OUT = 16 + 16 + 16 + 4  # A, B, T of the surviving path + next-queue id
SHADE_PENDING_OUT = 16 + 4          # CE (emission|flags) + connect-queue id
SHADE_RAY_OUT = 48                  # contribution|gid, origin|dir.x, dir.yz per queued shadow ray
SHADE_SCENE = 8 + 48 + 36 + 12 + 96 + 112 + 5 * 4 + (8 + 64) + (80 + 12 + 96 + 12 + 4) + 2 * 16  # instance, indices, 3 vertices, material, 5 1x1 texels, env alias + 4 texels, light entry + triangle, 2 LUT taps
CONNECT_FIXED = 4 + 16 + 16 + 16 + 16   # queue id, CE, T (pre-update throughput), pathLight read + write
CONNECT_RAY = 48                    # per shadow ray: the three records shade queued
CONNECT_FINAL = 16                  # frame-sum write at the end of a sample (samples_per_frame == 1)
PRIMARY_DONE = 16                   # a path that ends at bounce 0 writes only its frame sum
PRIMARY_ALIVE = 16 * 4 + 4          # a survivor writes records A, B, T, L and its queue id
RESOLVE_BYTES = 16 + 32             # per (pixel, frame) sum in, plus image read+write (amortised over the frames of a batch)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="frames per step per GPU (0 = backend default, ~4M resident paths)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the oracle sample")
    return ap.parse_args()


def cpu_baseline(vpt, scene, seconds):
    """Oracle on the host cores: bounded sample of the same workload (whole 1080p frames, depth 8)."""
    from oracle import oracle_py
    cores = os.cpu_count() or 1
    o = oracle_py.Oracle(scene, WIDTH, HEIGHT, threads=cores)
    o.set_params(vpt.default_params(max_depth=MAX_DEPTH, base_seed=BASE_SEED))
    t0 = time.perf_counter()
    o.render(1)
    t1 = time.perf_counter() - t0
    frames = max(1, min(32, int(seconds / max(t1, 1e-3)) - 1))
    t0 = time.perf_counter()
    o.render(frames)
    dt = time.perf_counter() - t0
    c = o.counters()
    o.close()
    return {"value": round(WIDTH * HEIGHT * frames / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d full 1920x1080 frames (1 spp each, depth %d) of the same Cornell workload, OpenMP over rows" % (frames, MAX_DEPTH),
            "mrays_per_s": round((c["closest"] + c["shadow"]) * frames / (frames + 1) / dt / 1e6, 3)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the backend has no CPU fallback")
    # Test hooks for a 1-GPU box (the N > 1 path is otherwise only ever run by the driver): VPT_BENCH_DEVICE pins every
    # rank to one device, VPT_BENCH_BACKEND=gloo routes the two collectives through host memory.  Never set by default.
    backend = os.environ.get("VPT_BENCH_BACKEND", "nccl")
    if "VPT_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["VPT_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    cdev = "cuda" if backend == "nccl" else "cpu"  # where collective operands live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    vpt = importlib.import_module("vulkan-path-tracer_amd")
    sharding = importlib.import_module("vulkan-path-tracer_amd.sharding")
    scene = vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz"))
    params = vpt.default_params(max_depth=MAX_DEPTH, base_seed=BASE_SEED, max_samples=0x7fffffff)

    # traversal visit counts (algorithmic bytes of extend / shadow) from a short counting pass
    cnt = vpt.PathTracer(WIDTH, HEIGHT, device=local_rank, shard_rank=rank, shard_count=world, count_traversal=True)
    cnt.set_scene(scene); cnt.set_params(params); cnt.render(2)
    cs = cnt.stats(); cnt.close()
    nodes_per_ray = cs["nodes_visited"] / max(cs["closest_rays"], 1)
    tris_per_ray = cs["tris_tested"] / max(cs["closest_rays"], 1)
    snodes_per_ray = cs["shadow_nodes_visited"] / max(cs["shadow_rays"], 1)
    stris_per_ray = cs["shadow_tris_tested"] / max(cs["shadow_rays"], 1)

    pt = vpt.PathTracer(WIDTH, HEIGHT, device=local_rank, shard_rank=rank, shard_count=world, profile=True,
                        frames_in_flight=args.frames_in_flight)
    pt.set_scene(scene); pt.set_params(params)
    F = pt.stats()["frames_in_flight"]
    shard = torch.empty(pt.shard_floats(), dtype=torch.float32, device="cuda")

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        pt.render(F)
    def gather():
        pt.shard_to_device(shard.data_ptr())
        return sharding.gather_shards(shard.to(cdev), world).to("cuda")

    if world > 1:  # warm the communicator outside the timed region
        gather()
    pt.reset_stats()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pt.render(F)
    gathered = gather()
    if rank == 0:
        pt.assemble_shards(gathered.data_ptr(), world)
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    st = pt.stats()
    local_samples = st["samples"]
    tot = torch.tensor([float(local_samples), float(st["closest_rays"]), float(st["shadow_rays"])], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    samples, closest, shadow = (float(x) for x in tot.tolist())

    if rank == 0:
        # ---- per-kernel algorithmic bytes / measured HIP-event time (this rank's launches)
        n0 = st["samples"] // 1                # bounce 0 of every slot runs in the fused primary kernel
        n_later = st["closest_rays"] - n0      # path-bounces that went through extend / shade
        hits0, alive0, rays0 = st["primary_hits"], st["primary_survivors"], st["primary_shadow_rays"]
        later_rays = st["shadow_rays"] - rays0
        alive_later = max(n_later - alive0, 0)  # paths leaving bounce k >= 1 alive == paths entering bounce k+1
        NODE_BYTES = st["bvh_node_bytes"]
        trav = nodes_per_ray * NODE_BYTES + tris_per_ray * TRI_BYTES
        strav = snodes_per_ray * NODE_BYTES + stris_per_ray * TRI_BYTES
        # per stage: (units, algorithmic bytes per unit by SURVEY 8d's formula = path records + scene gathers + BVH visits,
        #             of which bytes per unit that are path records / queues / frame sums, i.e. unique per path and bound for HBM)
        scene_hit = SHADE_SCENE
        units = {
            # fused bounce 0, per slot: frame sum out for paths that end, records A,B,T,L + queue id for survivors
            "primary": (n0, (PRIMARY_DONE * (n0 - alive0) + PRIMARY_ALIVE * alive0 + scene_hit * hits0 + strav * rays0) / max(n0, 1) + trav,
                        (PRIMARY_DONE * (n0 - alive0) + PRIMARY_ALIVE * alive0) / max(n0, 1)),
            # fused later bounce: records A,B,T,L in, the same out for survivors, frame sum for paths that end
            "bounce": (n_later, 4 + 64 + scene_hit + trav + (PRIMARY_ALIVE * alive_later + PRIMARY_DONE * alive0 + strav * later_rays) / max(n_later, 1),
                       4 + 64 + (PRIMARY_ALIVE * alive_later + PRIMARY_DONE * alive0) / max(n_later, 1)),
            "extend": (n_later, EXTEND_FIXED + trav, EXTEND_FIXED),
            "shade": (n_later, SHADE_IN + scene_hit + (SHADE_ALIVE_OUT * alive_later + SHADE_PENDING_OUT * st["connect_paths"] + SHADE_RAY_OUT * later_rays) / max(n_later, 1),
                      SHADE_IN + (SHADE_ALIVE_OUT * alive_later + SHADE_PENDING_OUT * st["connect_paths"] + SHADE_RAY_OUT * later_rays) / max(n_later, 1)),
            "connect": (st["connect_paths"], CONNECT_FIXED + (CONNECT_FINAL * alive0 + (CONNECT_RAY + strav) * later_rays) / max(st["connect_paths"], 1),
                        CONNECT_FIXED + (CONNECT_FINAL * alive0 + CONNECT_RAY * later_rays) / max(st["connect_paths"], 1)),
            "resolve": (st["samples"], RESOLVE_BYTES, RESOLVE_BYTES),
        }
        kernels = {}
        for name, (n, bpu, spu) in units.items():
            ms, launches = st["kernel_ms"][name], st["kernel_launches"][name]
            if launches == 0 or ms <= 0:
                continue
            kernels[name] = {"launches": launches, "avg_ms": round(ms / launches, 5), "share": 0.0,
                             "bytes_per_unit": round(bpu, 1), "record_bytes_per_unit": round(spu, 1), "units_per_launch": round(n / launches, 1),
                             "achieved_GBs": round(n * bpu / (ms * 1e-3) / 1e9, 2), "achieved_records_only_GBs": round(n * spu / (ms * 1e-3) / 1e9, 2)}
        tot_ms = sum(st["kernel_ms"][k] for k in kernels)
        for k in kernels:
            kernels[k]["share"] = round(st["kernel_ms"][k] / tot_ms, 4)
        dom = max(kernels, key=lambda k: kernels[k]["share"])
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")   # written from rocprofv3 --pmc passes (see profiles/README.md)
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roof = {"bound": "hbm", "kernel": dom, "achieved": kernels[dom]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(kernels[dom]["achieved_GBs"] / HBM_PEAK_GBS, 5), "traffic": traffic,
                "achieved_records_only": kernels[dom]["achieved_records_only_GBs"],
                "note": "achieved counts scene/BVH gathers that this 12-triangle scene serves from LDS/L1; records_only is the part that must cross HBM; traffic is the PMC measurement",
                "avg_launch_ms": kernels[dom]["avg_ms"],
                "algorithmic_bytes_per_launch": round(kernels[dom]["bytes_per_unit"] * kernels[dom]["units_per_launch"], 0),
                "traversal": {"nodes_per_closest_ray": round(nodes_per_ray, 3), "tris_per_closest_ray": round(tris_per_ray, 3),
                              "nodes_per_shadow_ray": round(snodes_per_ray, 3), "tris_per_shadow_ray": round(stris_per_ray, 3)},
                "kernels": kernels}
        line = {
            "metric": "Msamples/s at 1920x1080", "value": round(samples / dt / 1e6, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cornell_1080p_d8", "scene": "CornellBox (12 triangles, emissive quad 50), black env",
                       "width": WIDTH, "height": HEIGHT, "max_depth": MAX_DEPTH, "samples_per_frame": 1,
                       "frames_per_step_per_gpu": F, "paths_in_flight_per_gpu": st["shard_pixels"] * F,
                       "partition": "rows y % N == rank, one gather at the end", "base_seed": BASE_SEED},
            "mrays_per_s": round((closest + shadow) / dt / 1e6, 2),
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(vpt, scene, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    pt.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
