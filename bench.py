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

Launching: `python bench.py --gpus N` is enough.  Without a launcher environment (no WORLD_SIZE) and N > 1 the script starts
its own `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` and relays rank 0's JSON line;
under torchrun / the driver's launcher it is one rank.  Fewer visible devices than ranks is an error (exit 2) — unless
VPT_BENCH_DEVICE pins every rank to one device, the 1-GPU-box test hook (the gather then goes through host memory).

After the timed region the same workload runs a few more steps with profiling ON (HIP events the library records
around every launch on its own stream) and with traversal counters, which feed

  roofline     — for the kernel with the largest share of GPU time (the headline's is the whole-path launch k_whole,
                 timed under "primary"): `bound` says what limits it ("valu" for the Cornell kernels: VALU issue
                 saturated in profiles/, their BVH rides in LDS; "hbm" only where the bytes really cross HBM).  `achieved` = bytes that MUST cross HBM per launch (path records, queue
                 words, frame sums) / mean launch duration, so frac <= 1 by construction; `traffic`, `valu_busy` and the lane
                 use of the headline's kernel are MEASURED BY THIS RUN at N = 1 (live_pmc: three short child runs of this script
                 under `rocprofv3 --pmc`, one counter group each; `roofline.pmc` says so and keeps the committed file's figures
                 beside them), for the headline and for every extra workload's dominant kernel; where rocprofv3 is unavailable they
                 come from profiles/traffic.json under its source-id rule; `algorithmic_GBs` is SURVEY §8d's figure (records + scene
                 gathers + measured BVH visits), which on an LDS/L2-resident scene exceeds what HBM moves and is
                 therefore reported beside the fraction, not as it.
  workloads    — the scenes whose traversal touches memory, BASELINE configs 3, 4 and 5 at their own resolution and
                 depth ("atrium_1080p_d8", "atrium_4k_d8" = 3840x2160, "glass_bust_1080p_d32" with bloom + tonemap
                 on the root INSIDE its timed region): Msamples/s and, from rank 0, per-kernel algorithmic bytes from
                 MEASURED node / triangle visits with the traversal kernels' fraction of the 8 TB/s roofline.  At N > 1
                 every workload runs sharded over all ranks, once weak (per-rank batch fixed) and once strong (job fixed).
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
BASE_SEED = 1
RESIDENT_PATHS = 448 << 20   # what a context keeps in flight by default (vpt_api.hip kResidentPaths): the N = 1 batch is RESIDENT_PATHS / pixels frames
# name -> width, height, max depth, frames per step of the STRONG-scaling job (the fixed job is steps x this many whole frames), post, scene
WORKLOADS = {
    "cornell_1080p_d8": {"w": 1920, "h": 1080, "depth": 8, "strong_frames": 129, "post": False,      # 8 steps x 129 = 1032 spp: config 2's 1024-spp job
                         "scene": "CornellBox (12 triangles, emissive quad 50), black env"},
    "atrium_1080p_d8": {"w": 1920, "h": 1080, "depth": 8, "strong_frames": 32, "post": False,        # 8 steps x 32 = 256 spp: config 3's job
                        "scene": "procedural Sponza-class atrium, 253,002 triangles, 25 PBR materials, 12 textures, sun-and-sky env 2048x1024"},
    "atrium_4k_d8": {"w": 3840, "h": 2160, "depth": 8, "strong_frames": 32, "post": False,           # config 4's scene and resolution (its 1024 spp = 32 such steps)
                     "scene": "the same atrium at 3840x2160 (BASELINE config 4: pixel rows sharded over the GPUs, one RCCL gather)"},
    "glass_bust_1080p_d32": {"w": 1920, "h": 1080, "depth": 32, "strong_frames": 129, "post": True,  # config 5: bloom + tonemap on the root inside the timed region
                             "scene": "glass bust 510,992 triangles (transmission 1, roughness 0.05, IOR 1.5) on a plinth, sun-and-sky env 4096x2048; bloom + tonemap on the gathered image"},
}
EXTRA_WORKLOADS = ("atrium_1080p_d8", "atrium_4k_d8", "glass_bust_1080p_d32")

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
WHOLE_FINAL = 16                    # whole-path launch: a sample's frame sum, written once
WHOLE_PARKED = 32                   # ... and per hit of a bounce >= 1: pathLight out to the frame-sum slot and back (16 + 16)
# resolve: 16 B frame sum per (pixel, frame) in, plus one 32 B image read+write per pixel per batch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)   # weak: 8 batches of 226 frames = 1808 samples per pixel (covers config 2's 1024 spp); strong: 8 x 129 frames = its 1032-spp job
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cornell_1080p_d8", choices=sorted(WORKLOADS), help="headline workload (the default is BASELINE's metric config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-workloads", action="store_true", help="skip the atrium / atrium 4K / glass-bust blocks")
    ap.add_argument("--extra-steps", type=int, default=4, help="timed steps of each extra workload (a step of a regenerating context is a batch of 4 x 226 frames at 1080p)")
    ap.add_argument("--frames-in-flight", type=int, default=0, help="frames per step per GPU (0 = backend default, ~448M resident paths)")
    ap.add_argument("--pipeline", type=int, default=0, help="vpt_config.pipeline (0 AUTO)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the oracle sample")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every rank renders `steps` batches of its own default size (~448M resident paths), so work grows with N; "
                         "strong: the FIXED job steps x strong_frames whole frames (Cornell: 8 x 129 = config 2's 1024-spp job) is split over the ranks")
    ap.add_argument("--no-latency", action="store_true", help="skip the per-frame latency block (the reference's per-frame call pattern)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not spawn the rocprofv3 --pmc passes behind roofline.traffic / valu_busy (then they come from profiles/traffic.json under its staleness rule)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)   # live_pmc's child runs: the timed loop only, no line
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with no launcher environment: become the launcher.  One rank per GPU under torch.distributed.run on
    127.0.0.1, the same command line; rank 0's JSON line comes back on this process's stdout.  Refuses (exit 2) when the box shows
    fewer devices than ranks — a run that silently timed one GPU would be a void measurement."""
    import socket
    import subprocess
    one_device = "VPT_BENCH_DEVICE" in os.environ
    if not one_device:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d but %d HIP device(s) visible (VPT_BENCH_DEVICE=<ordinal> is the one-device test hook)\n" % (args.gpus, have))
            raise SystemExit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))   # torchrun would pin it to 1: the atrium's host-side BVH build and the oracle use threads
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def load_scene(vpt, name):
    if name == "cornell_1080p_d8":
        return vpt.scenes.Scene.load(os.path.join(ROOT, "tests", "golden", "cornell_box.npz"))
    if name.startswith("atrium"):
        return vpt.scenes.atrium()
    return vpt.scenes.glass_bust()


def cpu_baseline(vpt, name, scene, seconds):
    """Oracle on the host cores: bounded sample of the same workload (whole frames at the workload's resolution)."""
    from oracle import oracle_py
    W, H, depth = WORKLOADS[name]["w"], WORKLOADS[name]["h"], WORKLOADS[name]["depth"]
    cores = os.cpu_count() or 1
    o = oracle_py.Oracle(scene, W, H, threads=cores)
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
    return {"value": round(W * H * frames / dt / 1e6, 4), "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": "%d full %dx%d frames (1 spp each, depth %d) of the same workload, OpenMP over rows" % (frames, W, H, depth),
            "mrays_per_s": round((c["closest"] + c["shadow"]) * frames / (frames + 1) / dt / 1e6, 3)}


def traversal_counts(vpt, name, scene, params, device, rank, world, pipeline, frames):
    """Mean BVH nodes / triangles visited per closest-hit and per shadow ray, from the counting kernel variants."""
    cnt = vpt.PathTracer(WORKLOADS[name]["w"], WORKLOADS[name]["h"], device=device, shard_rank=rank, shard_count=world, count_traversal=True, pipeline=pipeline,
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
    whole = st["kernel_launches"]["bounce"] == 0 and st["kernel_launches"]["extend"] == 0 and st["kernel_launches"]["primary"] > 0   # one whole-path launch per batch (k_whole)
    streams = st["kernel_launches"]["join"] > 0                                            # staged pipeline on compact streams
    # bounce 0 ran in the fused primary kernel.  NOT on the streams: there "bounce" is the one-launch finisher (k_finish, timed under VPT_K_BOUNCE),
    # which round 5's table mistook for fused bounces and so priced the raygen launch and the finisher with the fused kernels' units (non-physical rates)
    fused0 = not streams and (st["kernel_launches"]["bounce"] > 0 or st["kernel_launches"]["extend"] == 0)
    fin_paths, fin_closest, fin_shadow = st.get("finish_paths", 0), st.get("finish_closest_rays", 0), st.get("finish_shadow_rays", 0)
    n_later = st["closest_rays"] - (n0 if fused0 else 0) - (fin_closest if streams else 0)   # closest-hit rays of the stream stages
    hits0, alive0, rays0 = st["primary_hits"], st["primary_survivors"], st["primary_shadow_rays"]
    later_rays = st["shadow_rays"] - rays0 - (fin_shadow if streams else 0)
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
    if whole:   # kernels_path.hip k_whole: the unit is a SAMPLE (camera ray to the path's end).  What must cross HBM per sample: its 16 B frame sum, and for
        # every hit of a bounce >= 1 (vpt_stats.connect_paths) pathLight's round trip through the frame-sum slot while the hit is parked: 16 B out + 16 B back
        rec = WHOLE_FINAL + WHOLE_PARKED * pend / max(n0, 1)
        units["primary"] = (n0, rec + (SHADE_SCENE * (hits0 + pend) + trav * st["closest_rays"] + strav * st["shadow_rays"]) / max(n0, 1), rec)
    if streams:
        rec = SHADE_STREAM_IN + (SHADE_STREAM_ALIVE * alive_later + SHADE_STREAM_PENDING * pend + 32 * later_rays) / max(n_later, 1)
        units["shade"] = (n_later, rec + SHADE_SCENE, rec)
        units["shadow"] = (later_rays, SHADOW_RAY + strav, SHADOW_RAY)
        units["join"] = (pend, JOIN_FIXED + (CONNECT_FINAL * fin) / max(pend, 1), JOIN_FIXED + (CONNECT_FINAL * fin) / max(pend, 1))
        # k_finish: the unit is a PATH-BOUNCE it ran (one closest-hit ray each).  Records cross HBM once per path taken over (queue word + RA, RB, RT, RL in,
        # the frame sum out); per bounce the scene gathers and the BVH visits of its closest-hit ray and its share of the shadow rays
        fin_rec = (4 + 64 + CONNECT_FINAL) * fin_paths / max(fin_closest, 1)
        units["bounce"] = (fin_closest, fin_rec + SHADE_SCENE + trav + strav * fin_shadow / max(fin_closest, 1), fin_rec)
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


class Ranks:
    """The launcher's view of this process: rank, world, device and the control-plane helpers every timed region uses."""

    def __init__(self, rank, world, device, one_device, torch, dist):
        self.rank, self.world, self.device, self.one_device, self.torch, self.dist = rank, world, device, one_device, torch, dist

    def sync(self):   # barrier + device synchronisation: both sides of every timed region
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max(self, v):
        t = self.torch.tensor([v], dtype=self.torch.float64)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sums(self, vals):
        t = self.torch.tensor([float(v) for v in vals], dtype=self.torch.float64)
        if self.world > 1:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(x) for x in t.tolist()]


def timed_run(vpt, sharding, R, name, scene, pipeline, frames_in_flight, steps, warmup, scaling):
    """One workload on all ranks: W warm-up steps, then the timed region — barrier + sync, `steps` batches per rank (weak) or the fixed job
    split over the ranks (strong), the library's one gather of the row shards (N > 1), bloom + tonemap on the root for the workloads
    that have post, barrier + sync — and the max over ranks of its wall time.  Rows y % N == rank (RayGen.slang:16-25 along one axis)."""
    wl = WORKLOADS[name]
    params = vpt.default_params(max_depth=wl["depth"], base_seed=BASE_SEED, max_samples=0x7fffffff)
    pt = vpt.PathTracer(wl["w"], wl["h"], device=R.device, shard_rank=R.rank, shard_count=R.world, pipeline=pipeline, frames_in_flight=frames_in_flight)
    t_scene = time.perf_counter()
    pt.set_scene(scene); pt.set_params(params)
    t_scene = time.perf_counter() - t_scene
    F = pt.stats()["frames_in_flight"]
    comm = sharding.ShardComm(pt, R.rank, R.world, host_staged=R.one_device) if R.world > 1 else None   # ncclCommInitRank inside libvpt_hip.so; the id travels over gloo
    batches = [F] * steps
    if scaling == "strong":   # the job is fixed: each rank renders all of its frames for its rows, in batches of its own frames-in-flight
        left, batches = steps * wl["strong_frames"], []
        while left > 0:
            batches.append(min(F, left)); left -= batches[-1]
    for _ in range(warmup):
        pt.render(batches[0])
    if comm:
        comm.gather_and_assemble()   # warm the communicator outside the timed region
    if wl["post"] and R.rank == 0:
        pt.postprocess()             # post buffers are allocated on first use
    pt.reset_stats()
    R.sync()
    t0 = time.perf_counter()
    for nb in batches:
        pt.render(nb)
    if comm:
        comm.gather_and_assemble()   # one ncclGather of the row shards to rank 0 + row re-interleave there
    post_ms = None
    if wl["post"] and R.rank == 0:   # config 5: PostProcessor::PostProcess on the whole image, on the root
        tp = time.perf_counter(); pt.postprocess(); post_ms = round((time.perf_counter() - tp) * 1e3, 3)
    R.sync()
    dt = R.max(time.perf_counter() - t0)
    st = pt.stats()
    samples, closest, shadow = R.sums([st["samples"], st["closest_rays"], st["shadow_rays"]])
    rccl = None
    if comm:   # what RCCL itself says about the communicator the gather ran on (vpt_comm_get_info); host-staged: versions only
        rccl = comm.info()
        if R.one_device:
            rccl.update({"nranks": R.world, "rank": R.rank, "transport": "host-staged through the torch process group (VPT_BENCH_DEVICE test hook: RCCL refuses two ranks on one device)"})
        comm.close()
    out = {"value": round(samples / dt / 1e6, 3), "unit": "Msamples/s", "scaling": scaling, "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4),
           "frames_per_step_per_gpu": F, "resident_frames_per_gpu": st["resident_frames"],
           # paths whose records live in HBM at a time: the whole batch, or (regeneration by refill) resident_frames of it; a whole-path context
           # (resident_frames 1: one frame of records allocated, none used) keeps its paths in registers — 3 waves per SIMD of them
           "paths_in_flight_per_gpu": st["shard_pixels"] * min(F, st["resident_frames"]) if st["resident_frames"] > 1 else "registers (whole-path launch)",
           "timed_samples_per_pixel": round(samples / (wl["w"] * wl["h"]), 1),
           "batches_per_gpu": batches if len(set(batches)) > 1 else "%d x %d frames" % (len(batches), batches[0]),
           "mrays_per_s": round((closest + shadow) / dt / 1e6, 2), "rays_per_sample": round((closest + shadow) / max(samples, 1), 3),
           "set_scene_s": round(t_scene, 3), "bvh_build_ms": round(st.get("bvh_build_ms", 0.0), 1), "set_scene_ms": round(st.get("set_scene_ms", 0.0), 1),
           "width": wl["w"], "height": wl["h"], "max_depth": wl["depth"], "scene": wl["scene"]}
    if post_ms is not None:
        out["post_ms_in_timed_region"] = post_ms
    if rccl:
        out["rccl"] = rccl
    pt.close()
    return out


def kernel_profile(vpt, name, scene, device, rank, world, pipeline, frames_in_flight, steps):
    """Profiled pass (HIP events the library records around every launch, on the stream it launches on) + traversal counts, this rank only."""
    wl = WORKLOADS[name]
    params = vpt.default_params(max_depth=wl["depth"], base_seed=BASE_SEED, max_samples=0x7fffffff)
    pp = vpt.PathTracer(wl["w"], wl["h"], device=device, shard_rank=rank, shard_count=world, pipeline=pipeline, frames_in_flight=frames_in_flight, profile=True)
    pp.set_scene(scene); pp.set_params(params)
    F = pp.stats()["frames_in_flight"]
    for _ in range(2):        # warm-up batches (buffers of the chosen pipeline are allocated on first use)
        pp.render(F)
    pp.reset_stats()
    for _ in range(steps):
        pp.render(F)
    st = pp.stats(); pp.close()
    whole = st["kernel_launches"]["bounce"] == 0 and st["kernel_launches"]["extend"] == 0
    used = 5 if whole else 2 if st["kernel_launches"]["extend"] > 0 else 1    # the pipeline AUTO settled on: count with the same one ("bounce" launches alone do not say fused: the streams' one-launch finisher k_finish is timed under that name)
    tc = traversal_counts(vpt, name, scene, params, device, rank, world, used if pipeline == 0 else pipeline, min(F, 4))
    kernels = kernel_table(st, tc)
    return {"pipeline": "whole paths (one launch per batch)" if whole else "fused" if used == 1 else ("staged (streams)" if st["kernel_launches"]["join"] > 0 else "staged (round-1 kernels)"), "bvh": {"nodes": st["bvh_nodes"], "triangles": st["bvh_triangles"], "node_bytes": st["bvh_node_bytes"], "tri_bytes": st["bvh_tri_bytes"]},
            "traversal": {k: round(v, 3) for k, v in tc.items() if k.endswith("_ray")}, "kernels": kernels}


def frame_latency(vpt, name, scene, params, device, frames=30):
    """The reference's per-frame call pattern (Editor::Draw: PathTrace once, PostProcess every frame — Editor.cpp:116,129) on a context
    that keeps a single frame in flight.  Two forms: `blocking` = vpt_render(ctx, 1) + vpt_postprocess, host wall clock per frame incl.
    both blocking syncs and the 8 MB RGBA8 read-back (what round 3 reported as frame_ms); `async` = what the reference actually does —
    PathTrace(cmd) and PostProcess(cmd) record and return, the output stays on the device (PathTracer.h:94-95) — through
    vpt_render_async + vpt_postprocess_device with one vpt_wait per frame on an EARLIER frame's ticket (two or three frames in flight,
    like a double- / triple-buffered swapchain): steady-state wall clock per frame."""
    W, H = WORKLOADS[name]["w"], WORKLOADS[name]["h"]
    g = vpt.PathTracer(W, H, device=device, frames_in_flight=1)
    g.set_scene(scene); g.set_params(params)
    for _ in range(5):
        g.render(1); g.postprocess()
    tr = tp = 0.0
    for _ in range(frames):
        t0 = time.perf_counter(); g.render(1); t1 = time.perf_counter(); g.postprocess(); t2 = time.perf_counter()
        tr += t1 - t0; tp += t2 - t1
    out = {"blocking_frame_ms": round((tr + tp) / frames * 1e3, 4), "render_1spp_ms": round(tr / frames * 1e3, 4), "postprocess_ms": round(tp / frames * 1e3, 4),
           "frames": frames, "what": "per 1-spp frame at %dx%d, 1 frame in flight; blocking: vpt_render(ctx, 1) + vpt_postprocess incl. syncs and the RGBA8 read-back; "
                                     "frame_ms: vpt_render_async + vpt_postprocess_device (RGBA8 stays on the device), steady state with two frames in flight (after issuing frame k the host waits for frame k - 1), frame_ms_3_in_flight: with three" % (W, H)}
    def async_loop(in_flight, n):
        for _ in range(8):
            g.render_async(1); g.postprocess_device()
        g.wait()
        t0 = time.perf_counter(); tickets = []
        for _ in range(n):
            g.render_async(1)
            tickets.append(g.postprocess_device())
            if len(tickets) >= in_flight:
                g.wait(tickets[-in_flight])   # the frame-in-flight fence: the host records frame k while frames k - in_flight + 1 .. k may still be on the device
        g.wait()
        return round((time.perf_counter() - t0) / n * 1e3, 4)
    out["frame_ms"] = async_loop(2, frames * 4)                 # two frames in flight (a double-buffered swapchain)
    out["frame_ms_3_in_flight"] = async_loop(3, frames * 4)     # three (triple buffering): all three lanes busy
    out["graph"] = bool(g.stats().get("graph_launches", 0))
    g.close()
    return out


# stage name of bench.py's kernel table -> the kernels rocprofv3 lists for it (the counting instantiations of kernel_profile's traversal pass are not the timed ones)
def stage_of_kernel(kernel_name):
    import re
    m = re.search(r"<([^>]*)>", kernel_name)
    a = [x.strip() for x in m.group(1).split(",")] if m else []
    if "k_whole" in kernel_name:        # <COUNT, STRICT, PLAIN>
        return None if a and a[0] == "true" else "primary"
    if "k_bounce" in kernel_name:       # <LDS_SCENE, COUNT, FIRST, ...>
        if len(a) >= 2 and a[1] == "true":
            return None
        return "primary" if len(a) >= 3 and a[2] == "true" else "bounce"
    for key, st in (("k_trace_vote", "extend"), ("k_shade_stream", "shade"), ("k_trace_shadow", "shadow"), ("k_join", "join"), ("k_resolve", "resolve"),
                    ("k_finish<", "bounce"), ("k_refill_stream", "primary"), ("k_raygen_stream", "primary")):
        if key in kernel_name:
            return st
    return None


_LIVE_PMC_OFF = ""   # set by the first failed pass: later workloads do not try again


def live_pmc(name, stage, pipeline, frames_in_flight, steps=2):
    """roofline.traffic / valu_busy measured in THIS run: three short rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; the VALU counters — separate
    passes, never combined with a trace) over `python bench.py --workload <name> --steps 2|4 --warmup 1 --pmc-child` (the timed loop of this script and nothing else), the counters of the stage's kernels averaged per launch.  The corrections are profiles/summarize_bench_r05.py's: HBM-side bytes =
    (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (KiB units, the fetch counter doubled per MI355X_MICROARCH.md's gfx950 note), VALU busy =
    SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) capped at 1, lanes = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU.
    Returns (entry, info); entry is None when rocprofv3 is missing or a pass fails (info says why) and the caller falls back to profiles/traffic.json."""
    import csv, glob, shutil, subprocess, tempfile
    global _LIVE_PMC_OFF
    if _LIVE_PMC_OFF:   # a pass of an earlier workload failed or timed out: the line must still go out within minutes
        return None, {"error": "skipped after an earlier failure: " + _LIVE_PMC_OFF}
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, {"error": "rocprofv3 not found"}
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        return None, {"error": "already running under a profiler"}
    passes = (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("valu", ["SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE"]))
    child = [sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", str(steps), "--warmup", "1", "--pipeline", str(pipeline), "--frames-in-flight", str(frames_in_flight),
             "--no-cpu-baseline", "--no-extra-workloads", "--no-latency", "--no-live-pmc", "--pmc-child"]
    tmp = tempfile.mkdtemp(prefix="vpt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    acc, cnt, names, t0 = {}, {}, set(), time.perf_counter()
    try:
        for tag, counters in passes:
            cmd = [exe, "--pmc"] + counters + ["--output-format", "csv", "-d", os.path.join(tmp, tag), "-o", tag, "--"] + child
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=150)
            files = glob.glob(os.path.join(tmp, tag, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                _LIVE_PMC_OFF = "pass %s of %s: rc %d, %d counter file(s)" % (tag, name, r.returncode, len(files))
                return None, {"error": "pass %s: rc %d, %d counter file(s)" % (tag, r.returncode, len(files)), "tail": r.stdout.decode(errors="replace")[-300:]}
            for f in files:
                for row in csv.DictReader(open(f)):
                    if stage_of_kernel(row["Kernel_Name"]) == stage:
                        c = row["Counter_Name"]
                        acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"]); cnt[c] = cnt.get(c, 0) + 1
                        names.add(row["Kernel_Name"].split("(")[0])
    except Exception as e:   # a timeout, an unreadable file: the line still goes out, with the committed counters
        _LIVE_PMC_OFF = "%s in the passes of %s" % (type(e).__name__, name)
        return None, {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    need = ("FETCH_SIZE", "WRITE_SIZE", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE")
    if any(cnt.get(c, 0) == 0 for c in need):
        return None, {"error": "no %s launches in the passes" % stage, "counters_seen": sorted(cnt)}
    per = lambda c: acc[c] / cnt[c]
    fetch, write = per("FETCH_SIZE") * 1024.0, per("WRITE_SIZE") * 1024.0
    busy = per("SQ_ACTIVE_INST_VALU") * 4.0 / (1024.0 * per("GRBM_GUI_ACTIVE") / 8.0)
    entry = {"hbm_bytes_per_launch": 2.0 * fetch + write, "fetch_bytes_raw": fetch, "write_bytes": write, "valu_busy": round(min(busy, 1.0), 3), "valu_busy_raw": round(busy, 3),
             "lanes_per_valu_instr": round(acc["SQ_THREAD_CYCLES_VALU"] / acc["SQ_ACTIVE_INST_VALU"], 1), "kernel_names": sorted(names)}
    info = {"source": "live: rocprofv3 --pmc passes spawned by this run (FETCH_SIZE | WRITE_SIZE | SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE), `%s` launches averaged" % stage,
            "launches_per_pass": cnt["FETCH_SIZE"], "seconds": round(time.perf_counter() - t0, 1),
            "corrections": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; valu_busy = min(1, SQ_ACTIVE_INST_VALU x 4 / (1024 x GRBM_GUI_ACTIVE / 8)); lanes = SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU"}
    return entry, info


def roofline_for(name, prof, live=None):
    """The roofline object of the bench line, for the kernel with the largest share of GPU time.  live = (entry, info) of live_pmc for that kernel."""
    kernels = prof["kernels"]
    dom = max(kernels, key=lambda k: kernels[k]["share"])
    k = kernels[dom]
    pmc_all = load_json(os.path.join(ROOT, "profiles", "traffic.json"))   # written from rocprofv3 --pmc passes (profiles/README.md)
    pmc = pmc_all.get(name, {})
    entry = pmc.get(dom, {})
    # the counters belong to the build they were collected on: copied only when traffic.json carries the id of THIS build's sources
    # (vulkan-path-tracer_amd/_build.py source_id: kernels, headers, flags) and the recorded mean launch time is within 4 % of this run's own (boxes of this pool differ by ~3 %; round 4's rule was 5 % on the time alone)
    try:
        source_id = importlib.import_module("vulkan-path-tracer_amd._build").source_id()
    except Exception:
        source_id = None
    same_build = source_id is not None and pmc_all.get("_source_id") == source_id
    pmc_ms = entry.get("mean_duration_us", 0.0) / 1e3
    stale = bool(entry) and not (same_build and 0.96 <= pmc_ms / max(k["avg_ms"], 1e-9) <= 1.04)
    if stale:
        entry = {}
    file_entry = entry
    pmc_note = {"source": "profiles/traffic.json", "recorded_avg_launch_ms": round(pmc_ms, 5) if pmc_ms else None, "stale": stale, "source_id": source_id,
                "recorded_source_id": pmc_all.get("_source_id"), "kernel_names": entry.get("kernel_names"),
                "rule": "copied only when traffic.json was collected on this build (source_id of kernels + headers + flags) and its mean launch time is within 4 % of this run's (box-to-box spread ~3 %)"}
    if live is not None and live[0] is not None:
        # measured by this run: traffic, VALU busy and lane use come from the live passes; the operation-mix figures (a separate seven-counter pass) stay the committed file's
        entry = dict(file_entry, **live[0])
        pmc_note = dict(live[1], source_id=source_id, kernel_names=live[0]["kernel_names"],
                        committed_file={"traffic": file_entry.get("hbm_bytes_per_launch"), "valu_busy": file_entry.get("valu_busy"), "lanes_per_valu_instr": file_entry.get("lanes_per_valu_instr"), "stale": stale})
    elif live is not None:
        pmc_note["live_failed"] = live[1]
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
            "pmc": pmc_note,
            # the roof that actually binds these kernels (SURVEY 8d's secondary figure): VALU lane throughput.  frac = share of issue
            # cycles with a VALU instruction (VALUBusy) x share of its 64 lanes that are active; the peak is 256 CUs x 4 SIMDs x 16
            # lanes x 2.4 GHz lane-operations per second, of which the 157 TFLOP/s fp32 figure counts 4 flops each (packed FMA)
            "valu": None if valu_busy is None or lanes is None else {
                "busy": valu_busy, "active_lanes_of_64": lanes, "frac": round(min(valu_busy, 1.0) * lanes / 64.0, 4),
                "lane_ops_per_s": round(min(valu_busy, 1.0) * lanes / 64.0 * VALU_LANE_OPS_PEAK, 1), "peak_lane_ops_per_s": VALU_LANE_OPS_PEAK,
                "fp32_tflops_if_every_op_were_an_fma": round(min(valu_busy, 1.0) * lanes / 64.0 * VALU_LANE_OPS_PEAK * 2 / 1e12, 2), "fp32_peak_tflops": 157.3,
                # measured, not derived (profiles/collect_bench_r04.sh `flops` pass: SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 x 64 lanes x lane use / launch time)
                "fp32_tflops_measured": entry.get("fp32_tflops"), "fp32_frac_of_peak": None if entry.get("fp32_tflops") is None else round(entry["fp32_tflops"] / 157.3, 4),
                "fp32_share_of_valu_instr": entry.get("fp32_share_of_valu_instr"),
                "note": "the traversal kernels' instruction mix is ~40 % fp32 arithmetic (profiles/r03_trace_isa_budget.md), none of it packed"},
            "record_bytes_per_launch": round(k["record_bytes_per_unit"] * k["units_per_launch"], 0),
            "algorithmic_bytes_per_launch": round(k["algorithmic_bytes_per_unit"] * k["units_per_launch"], 0),
            "algorithmic_GBs": k["algorithmic_GBs"], "algorithmic_frac_of_hbm_peak": k["algorithmic_frac_of_hbm_peak"],
            "note": "achieved = path records / queue words / frame sums (must cross HBM) per launch / mean launch time; algorithmic_* adds scene gathers and measured "
                    "BVH visits by SURVEY 8d's formula, which LDS / L1 / L2 serve on a resident scene; traffic and valu_busy are rocprofv3 PMC passes (profiles/)"
                    + ("; this kernel (k_whole: a batch as ONE launch of persistent waves) keeps a path in registers from bounce to bounce, so the bytes that must cross HBM are the "
                       "16 B frame sum per sample plus 32 B per parked later hit — 109 -> ~32 B per sample against round 3's per-bounce kernels: `frac` FELL because the kernel "
                       "moves less, while Msamples/s rose 13-17 %; the roof that binds it is VALU issue (`valu`)" if prof.get("pipeline", "").startswith("whole") else ""),
            "traversal": prof["traversal"], "kernels": kernels}


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)   # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks\n" % (args.gpus, world))
        raise SystemExit(2)
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
    elif torch.cuda.device_count() < world:
        sys.stderr.write("bench.py: %d ranks but %d HIP device(s) visible\n" % (world, torch.cuda.device_count()))
        raise SystemExit(2)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only: the data-path gather is the library's RCCL call
    vpt = importlib.import_module("vulkan-path-tracer_amd")
    sharding = importlib.import_module("vulkan-path-tracer_amd.sharding")
    R = Ranks(rank, world, local_rank, one_device, torch, dist)
    name = args.workload
    wl = WORKLOADS[name]
    scene = load_scene(vpt, name)
    params = vpt.default_params(max_depth=wl["depth"], base_seed=BASE_SEED, max_samples=0x7fffffff)

    head = timed_run(vpt, sharding, R, name, scene, args.pipeline, args.frames_in_flight, args.steps, args.warmup, args.scaling)
    if args.pmc_child:
        return

    line = None
    if rank == 0:
        prof = kernel_profile(vpt, name, scene, local_rank, rank, world, args.pipeline, args.frames_in_flight, 4)
        live = None
        if world == 1 and not args.no_live_pmc:   # the dominant kernel's counters, measured here and now (three short child runs of this script under rocprofv3 --pmc)
            live = live_pmc(name, max(prof["kernels"], key=lambda k: prof["kernels"][k]["share"]), args.pipeline, args.frames_in_flight, steps=4)   # (1 + 4 launches of the whole-path kernel per pass)
        line = {
            "metric": "Msamples/s at %dx%d" % (wl["w"], wl["h"]), "value": head["value"], "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": name, "scene": wl["scene"],
                       "width": wl["w"], "height": wl["h"], "max_depth": wl["depth"], "samples_per_frame": 1,
                       "frames_per_step_per_gpu": head["frames_per_step_per_gpu"], "paths_in_flight_per_gpu": head["paths_in_flight_per_gpu"],
                       "timed_samples_per_pixel": head["timed_samples_per_pixel"],
                       "note": "a rate metric: the timed region renders steps x frames_per_step frames (timed_samples_per_pixel; config 2 asks for 1024 spp, which the default 8 steps cover); bit-exact full-size runs: profiles/*_config*_full_parity.json",
                       "batches_per_gpu": head["batches_per_gpu"], "post_in_timed_region": wl["post"],
                       "partition": "rows y % N == rank, one ncclGather at the end", "base_seed": BASE_SEED, "pipeline": prof["pipeline"]},
            "mrays_per_s": head["mrays_per_s"],
            "set_scene": {"wall_s": head["set_scene_s"], "bvh_build_ms": head["bvh_build_ms"], "set_scene_ms": head["set_scene_ms"]},
            "roofline": roofline_for(name, prof, live),
        }
        if "post_ms_in_timed_region" in head:
            line["post_ms_in_timed_region"] = head["post_ms_in_timed_region"]
        if "rccl" in head:
            line["rccl"] = head["rccl"]
        if world == 1 and not args.no_latency and name == "cornell_1080p_d8":
            line["latency"] = frame_latency(vpt, name, scene, params, local_rank)
    if not args.no_extra_workloads:   # every rank takes part: the extra workloads run sharded like the headline one
        extra = {}
        for other in EXTRA_WORKLOADS:
            if other == name:
                continue
            sc2 = load_scene(vpt, other)
            modes = ("weak",) if world == 1 else ("weak", "strong")
            runs = {m: timed_run(vpt, sharding, R, other, sc2, 0, 0, args.extra_steps, 2, m) for m in modes}
            if rank == 0:
                w = dict(runs["weak"])
                if world > 1:
                    w["strong"] = {k: runs["strong"][k] for k in ("value", "unit", "ms_per_step", "batches_per_gpu", "timed_samples_per_pixel", "mrays_per_s")}
                w.update(kernel_profile(vpt, other, sc2, local_rank, rank, world, 0, 0, 3))
                if world == 1 and not args.no_latency:   # one frame per call on the real scenes too (Editor.cpp:116,129): the streams pipeline's ~7 launches per bounce
                    wl2 = WORKLOADS[other]
                    w["latency"] = frame_latency(vpt, other, sc2, vpt.default_params(max_depth=wl2["depth"], base_seed=BASE_SEED, max_samples=0x7fffffff), local_rank, frames=10)
                    w["latency"]["batch_ms_per_frame"] = round(wl2["w"] * wl2["h"] / (w["value"] * 1e3), 4)   # what a frame costs inside a full batch: the floor
                live2 = None
                if world == 1 and not args.no_live_pmc:
                    live2 = live_pmc(other, max(w["kernels"], key=lambda k: w["kernels"][k]["share"]), 0, 0)
                r = roofline_for(other, w, live2)
                w["roofline"] = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_frac_of_hbm_peak", "valu_busy", "avg_launch_ms", "algorithmic_GBs", "algorithmic_frac_of_hbm_peak", "pmc", "valu")}
                extra[other] = w
            del sc2
        if rank == 0:
            line["workloads"] = extra
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(vpt, name, scene, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
