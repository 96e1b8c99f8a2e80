"""Edge cases: tiny and ragged image sizes, a scene with no triangles, degenerate triangles, depth 1, very large and
very small scene scales (the quantised BVH boxes must stay conservative), post-processing of images too small for
a full bloom chain."""
import copy

import numpy as np
import pytest

from test_gpu_parity import assert_parity

pytestmark = pytest.mark.gpu


def both(vpt, oracle, sc, w, h, P, frames, **kw):
    o = oracle.Oracle(sc, w, h); o.set_params(P); o.render(frames)
    ref = o.radiance(); o.close()
    g = vpt.PathTracer(w, h, **kw); g.set_scene(sc); g.set_params(P); g.render(frames)
    img = g.radiance(); out8 = g.postprocess(); g.close()
    ref8, _ = oracle.postprocess(ref, vpt.default_post_params())
    return img, ref, out8, ref8


@pytest.mark.parametrize("w,h", [(1, 1), (1, 7), (5, 3), (2, 2), (63, 65), (257, 3)])
@pytest.mark.parametrize("pipeline", [1, 2])
def test_tiny_and_ragged_image_sizes(vpt, oracle, scenes, w, h, pipeline):
    img, ref, out8, ref8 = both(vpt, oracle, scenes("cornell_box"), w, h, vpt.default_params(max_depth=5), 3, pipeline=pipeline, build_flags=4 if pipeline == 2 else 0)
    assert_parity(img, ref)
    assert np.array_equal(out8, ref8)       # bloom chain stops when a mip would fall below 2 px (PostProcessor.cpp:136-157)


def scaled(sc, k):
    s = copy.deepcopy(sc)
    s.instances = [(m, mat, (np.diag([k, k, k, 1.0]) @ x).astype(np.float32)) for m, mat, x in s.instances]
    v = np.linalg.inv(s.view_inverse.astype(np.float64)); cam = s.view_inverse.astype(np.float64).copy(); cam[:3, 3] *= k
    s.view_inverse = cam.astype(np.float32)
    return s


@pytest.mark.parametrize("k", [1e-3, 1e3, 3.7e4])
def test_scene_scale_does_not_break_the_quantised_bvh(vpt, oracle, scenes, k):
    """Random rays against the viking room scaled by k: closest hits equal the oracle's (its own BVH / brute force)."""
    sc = scaled(scenes("viking_room"), k)
    rng = np.random.default_rng(3)
    n = 60000
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = rng.uniform(-2 * k, 2 * k, (n, 3))
    d = rng.normal(size=(n, 3)); rays[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    rays[:, 3] = 1e-4 * k; rays[:, 7] = 1e6 * k
    o = oracle.Oracle(sc, 8, 8); ref = o.trace_rays(rays); o.close()
    g = vpt.PathTracer(8, 8); g.set_scene(sc); got = g.trace_rays(rays); g.close()
    for f in ("t", "u", "v", "primitive", "instance"):
        assert np.array_equal(got[f], ref[f]), f
    assert (ref["t"] >= 0).mean() > 0.02


def test_depth_one_and_degenerate_triangles(vpt, oracle, scenes):
    sc = copy.deepcopy(scenes("cornell_box"))
    S = vpt.scenes
    pos = np.array([[0, -5, 0], [0, -5, 0], [0, -5, 0],            # a point
                    [-1, -6, 0], [0, -6, 0], [1, -6, 0]], np.float32)  # a segment (collinear)
    m = sc.add_mesh(pos, np.tile([0, 0, 1], (6, 1)).astype(np.float32), np.zeros((6, 2), np.float32), np.arange(6, dtype=np.uint32))
    sc.add_instance(m, 0)
    for depth in (1, 2):
        img, ref, _, _ = both(vpt, oracle, sc, 96, 54, vpt.default_params(max_depth=depth), 2)
        assert_parity(img, ref)


def test_scene_without_triangles(vpt, oracle):
    """Only an environment: every ray misses; with the env shown directly the image is the env, bit for bit."""
    S = vpt.scenes
    sc = S.Scene()
    m = sc.add_mesh(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros((0, 2), np.float32), np.zeros(0, np.uint32))
    sc.materials.append(S.material()); sc.add_instance(m, 0)
    sc.env = S.sun_sky_env(64, 32, seed=1, sun_peak=100.0); sc.luts = S.load_luts()
    sc.view_inverse = np.linalg.inv(S.look_at((0, 0, 0), (0, -0.2, -1), (0, 1, 0))).astype(np.float32)
    img, ref, out8, ref8 = both(vpt, oracle, sc, 64, 36, vpt.default_params(max_depth=4), 2)
    assert_parity(img, ref)
    assert img[..., :3].max() > 0 and np.array_equal(out8, ref8)


@pytest.mark.parametrize("far", [30.0, 1000.0])
def test_rays_from_far_outside_the_scene(vpt, oracle, scenes, far):
    """Origins `far` scene radii away, aimed at the scene: the slab test's rounding grows with the distance to the box,
    the build-time padding does not — the traversal must still find every hit the shared triangle test accepts.
    (Holds up to ~1e3 scene radii.  At 3e4 radii about 0.1 % of the rays differ from brute force: there one ulp of the
    ray origin is larger than the triangles, and no fixed box padding can follow what the fp32 triangle test accepts;
    the integrator never starts a ray that far from the geometry it can hit: DESIGN.md §6.)"""
    sc = scenes("viking_room")
    rng = np.random.default_rng(11)
    n = 40000
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    target = rng.uniform(-0.8, 0.8, (n, 3))
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3] = target - d * far * 2.0
    rays[:, 4:7] = d
    rays[:, 3] = 1e-4; rays[:, 7] = 1e9
    o = oracle.Oracle(sc, 8, 8); o.set_brute_force(True); ref = o.trace_rays(rays); o.close()   # no box test on the oracle side at all
    g = vpt.PathTracer(8, 8); g.set_scene(sc); got = g.trace_rays(rays); g.close()
    for f in ("t", "u", "v", "primitive", "instance"):
        assert np.array_equal(got[f], ref[f]), (f, int((got[f] != ref[f]).sum()))
    assert (ref["t"] >= 0).mean() > 0.2
