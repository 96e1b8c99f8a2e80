"""The oracle's TraceRay shadow mode (VPT_FLAG_RAY_QUERIES clear; RTCommon.slang:64-84, 103-117, MissShadow.slang:4-9) on the CPU: what the mode
must and must not change, on scenes where that can be said without a second implementation."""
import copy

import numpy as np

from test_gpu_shadow_modes import far_ceiling_scene, no_rq, oracle_image


def test_tmax_1000_along_the_normalised_direction(vpt, oracle):
    """A ceiling 1500 units above the floor hides the sky from a ray query (TMax 1e6) and not from the TraceRay form (TMax 1000)."""
    sc = far_ceiling_scene(vpt)
    w, h = 64, 36
    a, _ = oracle_image(oracle, sc, w, h, no_rq(vpt, max_depth=2), 4)
    b, _ = oracle_image(oracle, sc, w, h, vpt.default_params(max_depth=2), 4)
    floor_a, floor_b = a[h * 3 // 4:, :, :3].mean(), b[h * 3 // 4:, :, :3].mean()
    assert floor_a > 1.5 * floor_b > 0.0, (floor_a, floor_b)   # (both see the sky below the ceiling's rim through BSDF-sampled directions; only the TraceRay form adds the sky NEE term)


def test_emissive_mesh_nee_never_counts_without_ray_queries(vpt, oracle, scenes):
    """The light-identity compare reads an undefined payload word upstream; pinned as never equal: clearing VPT_FLAG_MESH_MIS' visibility, not its
    draws.  So the image equals neither the ray-query one nor the one with mesh NEE switched off (whose random stream is four draws shorter per hit),
    and it is darker than the ray-query image on average (light arrives through BSDF-sampled hits of the lamp only, MIS-weighted)."""
    sc = scenes("cornell_box")
    w, h, frames = 64, 36, 16
    a, ca = oracle_image(oracle, sc, w, h, no_rq(vpt, max_depth=4), frames)
    b, cb = oracle_image(oracle, sc, w, h, vpt.default_params(max_depth=4), frames)
    p = no_rq(vpt, max_depth=4); p.flags &= ~vpt._abi.FLAG_MESH_MIS
    c, _ = oracle_image(oracle, sc, w, h, p, frames)
    assert not np.array_equal(a, b) and not np.array_equal(a, c)
    assert a[..., :3].mean() < 0.8 * b[..., :3].mean()
    assert ca["closest"] > 0 and cb["closest"] > 0


def test_sky_visibility_is_the_same_predicate_at_ordinary_distances(vpt, oracle, scenes):
    """With mesh NEE off on both sides only the query interval and the re-normalised direction differ: in a room a few units across the two modes
    must agree to rounding (a sky ray that starts 1e-5 above a surface meets no triangle between t = 1e-5 and 1e-4)."""
    sc = copy.deepcopy(scenes("cornell_box_glass"))
    sc.env = vpt.scenes.sun_sky_env(32, 16, seed=4, sun_peak=30.0)
    w, h, frames = 64, 36, 4
    pa = no_rq(vpt, max_depth=5); pa.flags &= ~vpt._abi.FLAG_MESH_MIS
    pb = vpt.default_params(max_depth=5); pb.flags &= ~vpt._abi.FLAG_MESH_MIS
    a, _ = oracle_image(oracle, sc, w, h, pa, frames)
    b, _ = oracle_image(oracle, sc, w, h, pb, frames)
    differing = (np.abs(a - b).max(axis=2) > 0).mean()
    assert differing < 0.01, differing
