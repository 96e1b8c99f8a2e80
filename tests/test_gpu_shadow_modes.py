"""SetUseRayQueries(false): the TraceRay forms of the shadow and distance queries (RTCommon.slang:64-84, 103-117, MissShadow.slang:4-9) —
normalised direction, TMin 1e-5, TMax 1000, accept-first-hit — on every pipeline, bit for bit against the oracle.  Upstream the light-identity
compare reads an undefined payload word in this mode; both sides pin it as "never equal" (include/vpt.h VPT_FLAG_RAY_QUERIES)."""
import copy
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_image(oracle, sc, w, h, params, frames, volumes=None):
    o = oracle.Oracle(sc, w, h)
    o.set_params(params)
    if volumes:
        o.set_volumes(volumes)
    o.render(frames)
    ref, ctr = o.radiance(), o.counters()
    o.close()
    return ref, ctr


def no_rq(vpt, **kw):
    return vpt.default_params(flags=vpt._abi.FLAGS_DEFAULT & ~vpt._abi.FLAG_RAY_QUERIES, **kw)


@pytest.mark.parametrize("name,depth", [("cornell_box", 8), ("cornell_box_glass", 12), ("viking_room", 6)])
@pytest.mark.parametrize("pipeline", [0, 1, 2, 4])
def test_trace_ray_shadow_mode_equals_the_oracle(vpt, oracle, scenes, name, depth, pipeline):
    sc = copy.deepcopy(scenes(name))
    sc.env = vpt.scenes.sun_sky_env(64, 32, seed=5, sun_peak=80.0)   # sky NEE rays on every scene (the Cornell boxes ship a black environment)
    w, h, frames = 128, 72, 3
    p = no_rq(vpt, max_depth=depth)
    ref, ctr = oracle_image(oracle, sc, w, h, p, frames)
    g = vpt.PathTracer(w, h, pipeline=pipeline, build_flags=4 if pipeline == 2 else 0)
    g.set_scene(sc); g.set_params(p); g.render(frames)
    img, st = g.radiance(), g.stats()
    g.close()
    assert np.array_equal(img, ref), (name, pipeline)
    assert st["closest_rays"] == ctr["closest"]
    # the mode is not a no-op: with ray queries the same scene gives another image (emissive-mesh NEE counts there)
    ref_rq, _ = oracle_image(oracle, sc, w, h, vpt.default_params(max_depth=depth), frames)
    if name != "viking_room":   # (no emissive mesh in the Viking room: only the query interval differs, which these rays do not notice)
        assert not np.array_equal(ref, ref_rq)


def far_ceiling_scene(vpt):
    """A 40 x 40 floor at y = 0 under a 8000 x 8000 ceiling 1500 units above it (the world is y-down: up = -y), uniform white sky, no lights: what
    the floor receives from the sky is decided by whether a sky ray's interval reaches the ceiling — TMax 1e6 (ray queries) does, TMax 1000 does not."""
    sc = vpt.scenes.Scene()
    sc.luts = vpt.scenes.load_luts()
    sc.materials = [vpt.scenes.material(name="floor", base_color=(0.8, 0.8, 0.8), roughness=1.0), vpt.scenes.material(name="ceiling", base_color=(0.5, 0.5, 0.5), roughness=1.0)]
    def quad(y, half, normal_y):
        pos = np.array([[-half, y, -half], [half, y, -half], [half, y, half], [-half, y, half]], np.float32)
        nrm = np.tile(np.array([0, normal_y, 0], np.float32), (4, 1))
        uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
        return sc.add_mesh(pos, nrm, uv, np.array([0, 1, 2, 0, 2, 3], np.uint32))
    sc.add_instance(quad(0.0, 20.0, -1.0), 0)
    sc.add_instance(quad(-1500.0, 4000.0, 1.0), 1)
    sc.env = vpt.scenes.constant_env((1, 1, 1), 16, 8)
    v = np.linalg.inv(vpt.scenes.look_at((0.0, -6.0, 14.0), (0.0, 0.0, 0.0), (0, -1, 0)))
    sc.view_inverse = (v @ np.diag([1.0, -1.0, 1.0, 1.0])).astype(np.float32)
    return sc


def test_sky_rays_longer_than_1000_units_miss_in_trace_ray_mode(vpt, oracle):
    sc = far_ceiling_scene(vpt)
    w, h, frames = 96, 54, 3
    means = []
    for params in (no_rq(vpt, max_depth=3), vpt.default_params(max_depth=3)):
        ref, _ = oracle_image(oracle, sc, w, h, params, frames)
        means.append(float(ref[h * 3 // 4, :, :3].mean()))   # rows that see the floor
        for pipeline in (1, 2):   # four triangles ride in LDS: the fused per-bounce kernel; and the stream kernels forced on the same tree in memory
            g = vpt.PathTracer(w, h, pipeline=pipeline, build_flags=4 if pipeline == 2 else 0)
            g.set_scene(sc); g.set_params(params); g.render(frames)
            assert np.array_equal(g.radiance(), ref), pipeline
            g.close()
    assert means[0] > 1.5 * means[1] > 0.0, means   # TMax 1000: sky NEE reaches the floor; TMax 1e6: the far ceiling hides it (BSDF-sampled directions see the sky below its rim in both)


@pytest.mark.parametrize("pipeline", [0, 1])
def test_distance_query_in_trace_ray_mode_with_fog(vpt, oracle, scenes, pipeline):
    """GetDistanceToGeometry without ray queries (RTCommon.slang:103-117): normalised direction, TMax 1000 — fused media kernel on the box that
    rides in LDS, media stages on the streams for the glass-sphere scene."""
    fog = vpt.volume(corner_min=(-2.5, 0.0, -2.5), corner_max=(2.5, 5.0, 2.5), color=(0.8, 0.8, 0.9), density=0.25)
    for name in ("cornell_box", "cornell_box_glass"):
        sc = copy.deepcopy(scenes(name))
        sc.env = vpt.scenes.sun_sky_env(32, 16, seed=2, sun_peak=40.0)
        w, h, frames = 96, 54, 2
        p = no_rq(vpt, max_depth=5)
        ref, _ = oracle_image(oracle, sc, w, h, p, frames, [fog])
        g = vpt.PathTracer(w, h, pipeline=pipeline, build_flags=4 if pipeline == 2 else 0)
        g.set_scene(sc); g.set_params(p); g.set_volumes([fog]); g.render(frames)
        assert np.array_equal(g.radiance(), ref), (name, pipeline)
        g.close()


def test_set_use_ray_queries_false_through_the_cpp_facade(vpt, oracle, tmp_path):
    """PathTracer::SetUseRayQueries(false) (PathTracer.cpp:1086-1096) through the C++ facade's CLI."""
    cli = os.path.join(ROOT, "vulkan-path-tracer_amd", "host", "vpt_render")
    luts = os.path.join(ROOT, "vulkan-path-tracer_amd", "assets", "lookup_tables.bin")
    gltf = os.path.join(ROOT, "tests", "golden", "cornell_box.gltf")
    rad, cam = str(tmp_path / "r.f32"), str(tmp_path / "c.f32")
    w, h, spp, depth = 128, 72, 4, 6
    out = json.loads(subprocess.check_output([cli, "--scene", gltf, "--luts", luts, "--size", "%dx%d" % (w, h), "--spp", str(spp), "--depth", str(depth),
                                              "--radiance", rad, "--camera", cam, "--no-ray-queries"]))
    assert out["samples"] == spp
    img = np.fromfile(rad, "<f4").reshape(h, w, 4)
    m = np.fromfile(cam, "<f4").reshape(2, 4, 4)
    sc = vpt.scenes.load_gltf(gltf)
    o = oracle.Oracle(sc, w, h)
    o.set_camera(m[0].T, m[1].T)
    o.set_params(no_rq(vpt, max_depth=depth, max_samples=spp))
    o.render(spp)
    ref = o.radiance(); o.close()
    assert np.array_equal(img, ref)
