#!/usr/bin/env python
"""bench.py — Msamples/s of the wavefront path tracer on BASELINE.json's metric config.

Headline workload (config.workload = "cornell_1080p_d8"): BASELINE.json configs[1] — Assets/CornellBox as shipped
(diffuse walls + emissive quad, strength 50), black environment, 1920x1080, max depth 8, 1 sample per pixel per
frame, base seed 1.  A *step* is one wavefront batch of the hot path: every rank renders
`frames_per_step` consecutive frames of its own rows (all bounces until the ray queue is empty, then resolve).
Rows are dealt round-robin over ranks, each rank keeps ~448M paths resident (226 frames of a whole 1080p image, 1808 frames of a 1/8 shard), so per-GPU work per step is fixed
(weak scaling) and

    value = (samples all ranks traced in the K timed steps) / (max over ranks of the wall time)

with scene, BVH, path state and accumulation image resident in HBM before the timed region, and with the library's
per-launch event profiling OFF.  The timed region ends with the single collective of the path: one RCCL gather of the
finished row shards issued by the library itself (vpt_comm_gather_shards, include/vpt.h) and the row re-interleave on
rank 0.  torch.distributed is only the launcher / control plane (rendezvous, barrier, max-over-ranks).

After the timed region the same workload runs a few more steps with profiling ON (HIP events the library records
around every launch on its own stream) and with traversal counters, which feed

  roofline     — for the kernel with the largest share of GPU time: `bound` says what limits it ("valu" for the fused
                 Cornell kernels: 94-97 % VALU-busy in profiles/, their BVH rides in LDS; "hbm" only where the
                 bytes really cross HBM).  `achieved` = bytes that MUST cross HBM per launch (path records, queue
                 words, frame sums) / mean launch duration, so frac <= 1 by construction; `traffic` = the PMC
                 measurement (profiles/traffic.json); `algorithmic_GBs` is SURVEY §8d's figure (records + scene
                 gathers + measured BVH visits), which on an LDS/L2-resident scene exceeds what HBM moves and is
                 therefore reported beside the fraction, not as it.
  workloads    — (N=1 only) the two scenes whose traversal touches memory, BASELINE configs 3 and 5 at their own
                 resolution and depth ("atrium_1080p_d8", "glass_bust_1080p_d32"): Msamples/s and per-kernel
                 algorithmic bytes from MEASURED node / triangle visits, with the traversal kernels' fraction of
                 the 8 TB/s roofline.
  cpu_baseline — the CPU oracle (oracle/, a port of the reference shaders; the reference has no CPU path) on this
                 box's host cores, on a bounded sample of the headline workload.
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
VALU_LANE_OPS_PEAK = 256 * 4 * 16 * 2.4e9   # lane-operations per second: CUs x SIMDs x lanes per cycle x clock
FRAMES_AT_N1 = 129      # the step of the strong-scaling job: 8 steps x 129 whole 1080p frames = 1032 spp, config 2's 1024-spp job (the weak-scaling
                        # step is the library's default batch, 226 frames at N=1)
WIDTH, HEIGHT, BASE_SEED = 1920, 1080, 1
WORKLOADS = {   # name -> (max depth, scene description)
    "cornell_1080p_d8": (8, "CornellBox (12 triangles, emissive quad 50), black env"),
    "atrium_1080p_d8": (8, "procedural Sponza-class atrium, 253,002 triangles, 25 PBR materials, 12 textures, sun-and-sky env 2048x1024"),
    "glass_bust_1080p_d32": (32, "glass bust 510,992 triangles (transmission 1, roughness 0.05, IOR 1.5) on a plinth, sun-and-sky env 4096x2048"),
}

# Algorithmic bytes per unit for each stage: state words actually read/written per path or
# ray by the algorithm with this build's struct sizes; BVH node/triangle visits are measured, not assumed.
TRI_BYTES = 48   # triangle record; the node size comes from vpt_stats (64 B quantised BVH4, 128 B fp32 when the BVH rides in LDS)
EXTEND_FIXED = 4 + 24 + 20          # queue id, origin+direction in, hit record out
SHADE_IN = 4 + 16 + 16 + 16 + 20    # queue id, records A (origin|rng), B (dir|depth), T (throughput|pdf), hit record
SHADE_ALIVE_OUT = 16 + 16 + 16 + 4  # A, B, T of the surviving path + next-queue id
SHADE_PENDING_OUT = 16 + 4          # CE (emission|flags) + connect-queue id
SHADE_RAY_OUT = 48                  # contribution|gid, origin|dir.x, dir.yz per queued shadow ray
SHADE_SCENE = 8 + 48 + 36 + 12 + 96 + 112 + 5 * 4 + (8 + 64) + (80 + 12 + 96 + 12 + 4) + 2 * 16  # instance, indices, 3 vertices, material, 5 texels, env alias + 4 texels, light entry + triangle, 2 LUT taps
CONNECT_FIXED = 4 + 16 + 16 + 16 + 16   # queue id, CE, T (pre-update throughput), pathLight read + write
CONNECT_RAY = 48                    # per shadow ray: the three records shade queued
CONNECT_FINAL = 16                  # frame-sum write at the end of a sample (samples_per_frame == 1)
PRIMARY_DONE = 16                   # a path that ends at bounce 0 writes only its frame sum
PRIMARY_ALIVE = 16 * 4 + 4          # a survivor writes records A, B, T, L and its queue id
# resolve: 16 B frame sum per (pixel, frame) in, plus one 32 B image read+write per pixel per batch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)   # weak: 8 batches of 226 frames = 1808 samples per pixel (covers config 2's 1024 spp); strong: 8 x 129 frames = its 1032-spp job
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cornell_1080p_d8", choices=sorted(WORKLOADS), help="headline workload (the default is BASELINE's metric config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-workloads", action="store_true", help="skip the atrium / glass-bust blocks")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="frames per step per GPU (0 = backend default, ~448M resident paths)")
    ap.add_argument("--pipeline", type=int, default=0, help="vpt_config.pipeline (0 AUTO)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the oracle sample")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every rank renders `steps` batches of its own default size (~256M resident paths), so work grows with N; "
                         "strong: the FIXED job steps x 129 frames of the whole 1080p image (8 steps = config 2's 1024-spp job) is split over the ranks")
    ap.add_argument("--no-latency", action="store_true", help="skip the per-frame latency block (vpt_render(1) + vpt_postprocess, the reference's call pattern)")
    return ap.parse_args()


def load_scene(vpt, name):
    if name == "cornell_1080p_d8":
        return vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz"))
    if name == "atrium_1080p_d8":
        return vpt.scenes.atrium()
    return vpt.scenes.glass_bust()


def cpu_baseline(vpt, scene, depth, seconds):
    """Oracle on the host cores: bounded sample of the same workload (whole 1080p frames)."""
    from oracle import oracle_py
    cores = os.cpu_count() or 1
    o = oracle_py.Oracle(scene, WIDTH, HEIGHT, threads=cores)
    o.set_params(vpt.default_params(max_depth=depth, base_seed=BASE_SEED))
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
            "sample": "%d full 1920x1080 frames (1 spp each, depth %d) of the same workload, OpenMP over rows" % (frames, depth),
            "mrays_per_s": round((c["closest"] + c["shadow"]) * frames / (frames + 1) / dt / 1e6, 3)}


def traversal_counts(vpt, scene, params, device, rank, world, pipeline, frames):
    """Mean BVH nodes / triangles visited per closest-hit and per shadow ray, from the counting kernel variants."""
    cnt = vpt.PathTracer(WIDTH, HEIGHT, device=device, shard_rank=rank, shard_count=world, count_traversal=True, pipeline=pipeline,
                         frames_in_flight=frames)
    cnt.set_scene(scene); cnt.set_params(params); cnt.render(frames)
    cs = cnt.stats(); cnt.close()
    return {"nodes_per_closest_ray": cs["nodes_visited"] / max(cs["closest_rays"], 1), "tris_per_closest_ray": cs["tris_tested"] / max(cs["closest_rays"], 1),
            "nodes_per_shadow_ray": cs["shadow_nodes_visited"] / max(cs["shadow_rays"], 1), "tris_per_shadow_ray": cs["shadow_tris_tested"] / max(cs["shadow_rays"], 1),
            "closest_rays": cs["closest_rays"], "shadow_rays": cs["shadow_rays"], "samples": cs["samples"]}


# stream pipeline (kernels_stream.hip): records in queue order, compact pending / shadow-ray streams
SHADE_STREAM_IN = 4 + 16 + 16 + 16 + 20     # queue entry (slot), records RA, RB, RT, hit record: coalesced
SHADE_STREAM_ALIVE = 4 + 48                 # survivor: queue entry + RA, RB, RT at its new position
SHADE_STREAM_PENDING = 64                   # pending record PE, PS, PL, PT
SHADOW_RAY = 32 + 1                         # ray record in, visibility byte out
JOIN_FIXED = 64 + 16 + 16                   # pending record in, pathLight read + write (by slot)


def kernel_table(st, tc):
    """Per stage: launches, mean ms, units, algorithmic bytes per unit (records + scene gathers + measured BVH visits) and
    the part of them that is unique per path (records / queue words / frame sums) and therefore has to cross HBM."""
    n0 = st["samples"]
    fused0 = st["kernel_launches"]["bounce"] > 0 or st["kernel_launches"]["extend"] == 0   # bounce 0 ran in the fused primary kernel
    streams = st["kernel_launches"]["join"] > 0                                            # staged pipeline on compact streams
    n_later = st["closest_rays"] - (n0 if fused0 else 0)
    hits0, alive0, rays0 = st["primary_hits"], st["primary_survivors"], st["primary_shadow_rays"]
    later_rays = st["shadow_rays"] - rays0
    alive_later = max(n_later - alive0, 0) if fused0 else max(st["closest_rays"] - n0, 0)
    node_b = st["bvh_node_bytes"]
    trav = tc["nodes_per_closest_ray"] * node_b + tc["tris_per_closest_ray"] * TRI_BYTES
    strav = tc["nodes_per_shadow_ray"] * node_b + tc["tris_per_shadow_ray"] * TRI_BYTES
    fin = alive0 if fused0 else n0     # frame-sum writes by the connect / join stage
    pend = st["connect_paths"]
    units = {
        "primary": (n0, (PRIMARY_DONE * (n0 - alive0) + PRIMARY_ALIVE * alive0 + SHADE_SCENE * hits0 + strav * rays0) / max(n0, 1) + trav,
                    (PRIMARY_DONE * (n0 - alive0) + PRIMARY_ALIVE * alive0) / max(n0, 1)) if fused0 else (n0, 52.0 + 16.0, 52.0 + 16.0),
        "bounce": (n_later, 4 + 64 + SHADE_SCENE + trav + (PRIMARY_ALIVE * alive_later + PRIMARY_DONE * alive0 + strav * later_rays) / max(n_later, 1),
                   4 + 64 + (PRIMARY_ALIVE * alive_later + PRIMARY_DONE * alive0) / max(n_later, 1)),
        "extend": (n_later, EXTEND_FIXED + trav, EXTEND_FIXED),
        "resolve": (st["samples"], 16 + 32.0 / max(st["frames_in_flight"], 1), 16 + 32.0 / max(st["frames_in_flight"], 1)),
    }
    if streams:
        rec = SHADE_STREAM_IN + (SHADE_STREAM_ALIVE * alive_later + SHADE_STREAM_PENDING * pend + 32 * later_rays) / max(n_later, 1)
        units["shade"] = (n_later, rec + SHADE_SCENE, rec)
        units["shadow"] = (later_rays, SHADOW_RAY + strav, SHADOW_RAY)
        units["join"] = (pend, JOIN_FIXED + (CONNECT_FINAL * fin) / max(pend, 1), JOIN_FIXED + (CONNECT_FINAL * fin) / max(pend, 1))
    else:
        units["shade"] = (n_later, SHADE_IN + SHADE_SCENE + (SHADE_ALIVE_OUT * alive_later + SHADE_PENDING_OUT * pend + SHADE_RAY_OUT * later_rays) / max(n_later, 1),
                          SHADE_IN + (SHADE_ALIVE_OUT * alive_later + SHADE_PENDING_OUT * pend + SHADE_RAY_OUT * later_rays) / max(n_later, 1))
        units["connect"] = (pend, CONNECT_FIXED + (CONNECT_FINAL * fin + (CONNECT_RAY + strav) * later_rays) / max(pend, 1),
                            CONNECT_FIXED + (CONNECT_FINAL * fin + CONNECT_RAY * later_rays) / max(pend, 1))
    kernels = {}
    for name, (n, bpu, spu) in units.items():
        ms, launches = st["kernel_ms"][name], st["kernel_launches"][name]
        if launches == 0 or ms <= 0:
            continue
        kernels[name] = {"launches": launches, "avg_ms": round(ms / launches, 5), "total_ms": round(ms, 3), "share": 0.0,
                         "units_per_launch": round(n / launches, 1), "algorithmic_bytes_per_unit": round(bpu, 1), "record_bytes_per_unit": round(spu, 1),
                         "algorithmic_GBs": round(n * bpu / (ms * 1e-3) / 1e9, 1), "records_GBs": round(n * spu / (ms * 1e-3) / 1e9, 1),
                         "algorithmic_frac_of_hbm_peak": round(n * bpu / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    tot_ms = sum(st["kernel_ms"][k] for k in kernels)
    for k in kernels:
        kernels[k]["share"] = round(st["kernel_ms"][k] / tot_ms, 4)
    return kernels


def load_json(path):
    try:
        return json.load(open(path))
    except Exception:
        return {}


def profile_workload(vpt, name, scene, device, rank, world, pipeline, frames_in_flight, steps, warmup):
    """Un-profiled throughput, then profiled per-kernel times, then traversal counts, for one workload on this rank."""
    depth = WORKLOADS[name][0]
    params = vpt.default_params(max_depth=depth, base_seed=BASE_SEED, max_samples=0x7fffffff)
    pt = vpt.PathTracer(WIDTH, HEIGHT, device=device, shard_rank=rank, shard_count=world, pipeline=pipeline, frames_in_flight=frames_in_flight)
    pt.set_scene(scene); pt.set_params(params)
    F = pt.stats()["frames_in_flight"]
    for _ in range(warmup):
        pt.render(F)
    pt.reset_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        pt.render(F)          # vpt_render returns with the stream drained
    dt = time.perf_counter() - t0
    st = pt.stats(); pt.close()
    out = {"value": round(st["samples"] / dt / 1e6, 2), "unit": "Msamples/s", "steps": steps, "frames_per_step": F, "ms_per_step": round(dt / steps * 1e3, 3),
           "mrays_per_s": round((st["closest_rays"] + st["shadow_rays"]) / dt / 1e6, 1), "scene": WORKLOADS[name][1], "max_depth": depth,
           "rays_per_sample": round((st["closest_rays"] + st["shadow_rays"]) / max(st["samples"], 1), 3)}
    out.update(kernel_profile(vpt, name, scene, device, rank, world, pipeline, frames_in_flight, max(2, min(steps, 4))))
    return out


def kernel_profile(vpt, name, scene, device, rank, world, pipeline, frames_in_flight, steps):
    depth = WORKLOADS[name][0]
    params = vpt.default_params(max_depth=depth, base_seed=BASE_SEED, max_samples=0x7fffffff)
    pp = vpt.PathTracer(WIDTH, HEIGHT, device=device, shard_rank=rank, shard_count=world, pipeline=pipeline, frames_in_flight=frames_in_flight, profile=True)
    pp.set_scene(scene); pp.set_params(params)
    F = pp.stats()["frames_in_flight"]
    for _ in range(2):        # warm-up batches (buffers of the chosen pipeline are allocated on first use)
        pp.render(F)
    pp.reset_stats()
    for _ in range(steps):
        pp.render(F)
    st = pp.stats(); pp.close()
    used = 1 if st["kernel_launches"]["bounce"] > 0 else 2    # the pipeline AUTO settled on: count with the same one
    tc = traversal_counts(vpt, scene, params, device, rank, world, used if pipeline == 0 else pipeline, min(F, 4))
    kernels = kernel_table(st, tc)
    return {"pipeline": "fused" if used == 1 else ("staged (streams)" if st["kernel_launches"]["join"] > 0 else "staged (round-1 kernels)"), "bvh": {"nodes": st["bvh_nodes"], "triangles": st["bvh_triangles"], "node_bytes": st["bvh_node_bytes"], "tri_bytes": st["bvh_tri_bytes"]},
            "traversal": {k: round(v, 3) for k, v in tc.items() if k.endswith("_ray")}, "kernels": kernels}


def frame_latency(vpt, scene, params, device, frames=30):
    """The reference's per-frame call pattern (Editor::Draw: PathTrace once, PostProcess every frame — Editor.cpp:116,129): one
    vpt_render(ctx, 1) followed by one vpt_postprocess per frame on a context that keeps a single frame in flight, wall clock per
    frame incl. both blocking syncs and the RGBA8 read-back."""
    g = vpt.PathTracer(WIDTH, HEIGHT, device=device, frames_in_flight=1)
    g.set_scene(scene); g.set_params(params)
    for _ in range(5):
        g.render(1); g.postprocess()
    tr = tp = 0.0
    for _ in range(frames):
        t0 = time.perf_counter(); g.render(1); t1 = time.perf_counter(); g.postprocess(); t2 = time.perf_counter()
        tr += t1 - t0; tp += t2 - t1
    g.close()
    return {"frame_ms": round((tr + tp) / frames * 1e3, 4), "render_1spp_ms": round(tr / frames * 1e3, 4), "postprocess_ms": round(tp / frames * 1e3, 4),
            "frames": frames, "what": "vpt_render(ctx, 1) + vpt_postprocess per frame at 1920x1080, 1 frame in flight, host wall clock incl. syncs and the 8 MB RGBA8 read-back"}


def roofline_for(name, prof):
    """The roofline object of the bench line, for the kernel with the largest share of GPU time."""
    kernels = prof["kernels"]
    dom = max(kernels, key=lambda k: kernels[k]["share"])
    k = kernels[dom]
    pmc = load_json(os.path.join(ROOT, "profiles", "traffic.json")).get(name, {})   # written from rocprofv3 --pmc passes (profiles/README.md)
    entry = pmc.get(dom, {})
    # the counters belong to the build they were collected on: an entry whose recorded mean launch time is not within 5 % of
    # this run's is a different kernel (or another clock) and is NOT copied into the line
    pmc_ms = entry.get("mean_duration_us", 0.0) / 1e3
    stale = bool(entry) and not (0.95 <= pmc_ms / max(k["avg_ms"], 1e-9) <= 1.05)
    if stale:
        entry = {}
    traffic = entry.get("hbm_bytes_per_launch")
    valu_busy = entry.get("valu_busy")
    lanes = entry.get("lanes_per_valu_instr")
    lds_scene = prof["bvh"]["node_bytes"] == 128
    # bound from the counters where profiles/traffic.json has them: whichever of VALU issue (rocprofv3 VALUBusy; a kernel that
    # saturates issue reads 0.93-1.0, profiles/r02_valu_calibration.md) and HBM-side traffic / peak is higher.  None of the
    # path kernels is a streaming HBM kernel, so "frac" below is what they leave of the memory roofline.
    traffic_frac = traffic / (k["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None
    if valu_busy is not None and traffic_frac is not None:
        bound = "valu" if valu_busy >= traffic_frac else "hbm"
    else:
        bound = "valu" if lds_scene or dom in ("extend", "connect", "shadow", "shade") else "hbm"
    return {"bound": bound, "kernel": dom, "achieved": k["records_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(k["records_GBs"] / HBM_PEAK_GBS, 5),
            "traffic": traffic, "traffic_frac_of_hbm_peak": round(traffic_frac, 4) if traffic_frac else None, "valu_busy": valu_busy, "avg_launch_ms": k["avg_ms"],
            "pmc": {"source": "profiles/traffic.json", "recorded_avg_launch_ms": round(pmc_ms, 5) if pmc_ms else None, "stale": stale,
                    "rule": "copied only when the recorded mean launch time is within 5 % of this run's"},
            # the roof that actually binds these kernels (SURVEY 8d's secondary figure): VALU lane throughput.  frac = share of issue
            # cycles with a VALU instruction (VALUBusy) x share of its 64 lanes that are active; the peak is 256 CUs x 4 SIMDs x 16
            # lanes x 2.4 GHz lane-operations per second, of which the 157 TFLOP/s fp32 figure counts 4 flops each (packed FMA)
            "valu": None if valu_busy is None or lanes is None else {
                "busy": valu_busy, "active_lanes_of_64": lanes, "frac": round(min(valu_busy, 1.0) * lanes / 64.0, 4),
                "lane_ops_per_s": round(min(valu_busy, 1.0) * lanes / 64.0 * VALU_LANE_OPS_PEAK, 1), "peak_lane_ops_per_s": VALU_LANE_OPS_PEAK,
                "fp32_tflops_if_every_op_were_an_fma": round(min(valu_busy, 1.0) * lanes / 64.0 * VALU_LANE_OPS_PEAK * 2 / 1e12, 2), "fp32_peak_tflops": 157.3,
                "note": "the traversal kernels' instruction mix is ~40 % fp32 arithmetic (profiles/r03_trace_isa_budget.md), none of it packed"},
            "record_bytes_per_launch": round(k["record_bytes_per_unit"] * k["units_per_launch"], 0),
            "algorithmic_bytes_per_launch": round(k["algorithmic_bytes_per_unit"] * k["units_per_launch"], 0),
            "algorithmic_GBs": k["algorithmic_GBs"], "algorithmic_frac_of_hbm_peak": k["algorithmic_frac_of_hbm_peak"],
            "note": "achieved = path records / queue words / frame sums (must cross HBM) per launch / mean launch time; algorithmic_* adds scene gathers and measured "
                    "BVH visits by SURVEY 8d's formula, which LDS / L1 / L2 serve on a resident scene; traffic and valu_busy are rocprofv3 PMC passes (profiles/)",
            "traversal": prof["traversal"], "kernels": kernels}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # this image's driver shares device memory across processes through dmabuf only (RCCL needs it)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the backend has no CPU fallback")
    # Test hook for a 1-GPU box (the N > 1 path is otherwise only ever run by the driver): VPT_BENCH_DEVICE pins every rank to
    # one device and the shard gather then goes through host memory (RCCL refuses two ranks on one device).  Never set by default.
    one_device = "VPT_BENCH_DEVICE" in os.environ
    if one_device:
        local_rank = int(os.environ["VPT_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only: the data-path gather is the library's RCCL call
    vpt = importlib.import_module("vulkan-path-tracer_amd")
    sharding = importlib.import_module("vulkan-path-tracer_amd.sharding")
    name = args.workload
    depth = WORKLOADS[name][0]
    scene = load_scene(vpt, name)
    params = vpt.default_params(max_depth=depth, base_seed=BASE_SEED, max_samples=0x7fffffff)

    pt = vpt.PathTracer(WIDTH, HEIGHT, device=local_rank, shard_rank=rank, shard_count=world, pipeline=args.pipeline, frames_in_flight=args.frames_in_flight)
    pt.set_scene(scene); pt.set_params(params)
    F = pt.stats()["frames_in_flight"]
    comm = None
    if world > 1:
        comm = sharding.ShardComm(pt, rank, world, host_staged=one_device)   # ncclCommInitRank inside libvpt_hip.so; the id travels over gloo

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # strong scaling: the job is fixed (steps x 129 whole frames) and each rank renders all of its frames for its rows, in batches of
    # its own frames-in-flight; weak (default): steps batches of the rank's own default size
    batches = [F] * args.steps
    if args.scaling == "strong":
        left, batches = args.steps * FRAMES_AT_N1, []
        while left > 0:
            batches.append(min(F, left)); left -= batches[-1]
    for _ in range(args.warmup):
        pt.render(batches[0])
    if comm:
        comm.gather_and_assemble()   # warm the communicator outside the timed region
    pt.reset_stats()
    sync()
    t0 = time.perf_counter()
    for nb in batches:
        pt.render(nb)
    if comm:
        comm.gather_and_assemble()   # one ncclGather of the row shards to rank 0 + row re-interleave there
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    st = pt.stats()
    tot = torch.tensor([float(st["samples"]), float(st["closest_rays"]), float(st["shadow_rays"])], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    samples, closest, shadow = (float(x) for x in tot.tolist())
    shard_pixels = st["shard_pixels"]
    comm_info = comm.info() if comm and not one_device else None
    pt.close()

    if rank == 0:
        prof = kernel_profile(vpt, name, scene, local_rank, rank, world, args.pipeline, args.frames_in_flight, 4)
        line = {
            "metric": "Msamples/s at 1920x1080", "value": round(samples / dt / 1e6, 3), "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": name, "scene": WORKLOADS[name][1],
                       "width": WIDTH, "height": HEIGHT, "max_depth": depth, "samples_per_frame": 1,
                       "frames_per_step_per_gpu": F, "paths_in_flight_per_gpu": shard_pixels * F,
                       "timed_samples_per_pixel": round(samples / (WIDTH * HEIGHT), 1),
                       "note": "a rate metric: the timed region renders steps x frames_per_step frames (timed_samples_per_pixel; config 2 asks for 1024 spp, which the default 8 steps cover); bit-exact full-size run: profiles/*_config2_full_parity.json",
                       "batches_per_gpu": batches if len(set(batches)) > 1 else "%d x %d frames" % (len(batches), batches[0]),
                       "partition": "rows y % N == rank, one ncclGather at the end", "base_seed": BASE_SEED, "pipeline": prof["pipeline"]},
            "mrays_per_s": round((closest + shadow) / dt / 1e6, 2),
            "roofline": roofline_for(name, prof),
        }
        if comm_info:   # what RCCL itself says about the communicator the gather ran on (vpt_comm_get_info)
            line["rccl"] = comm_info
        if world == 1 and not args.no_latency:
            line["latency"] = frame_latency(vpt, scene, params, local_rank)
        if world == 1 and not args.no_extra_workloads:
            extra = {}
            for other in ("atrium_1080p_d8", "glass_bust_1080p_d32"):
                if other == name:
                    continue
                sc2 = load_scene(vpt, other)
                w = profile_workload(vpt, other, sc2, local_rank, 0, 1, 0, 0, steps=6, warmup=5)
                r = roofline_for(other, w)
                w["roofline"] = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_frac_of_hbm_peak", "valu_busy", "avg_launch_ms", "algorithmic_GBs", "algorithmic_frac_of_hbm_peak", "pmc", "valu")}
                extra[other] = w
            line["workloads"] = extra
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(vpt, scene, depth, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        if comm:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
