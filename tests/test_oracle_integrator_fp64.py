"""The oracle's integrator held against tests/ref_integrator64.py, an independent float64 restatement of RayGen / ClosestHit / Miss /
Surface / Sampler written from the Slang sources: per-sample values of single pixels (orc_pixel_samples).
Everything the image is built from is in play: seeding and draw order, camera ray, surface frame with the default normal map,
emissive-triangle NEE with the light-identity shadow test, environment importance sampling through the alias map (built as
PathTracer.cpp:1161-1296 builds it) with its visibility test, the miss shader's environment lookup, VNDF + lobe sampling, EvaluateBSDF with energy compensation, MIS
weights of both strategies, textures (LINEAR / REPEAT taps, gamma, metallic-roughness maps), the luminance clamp, Russian roulette, refraction and the in-medium flag, the NaN guard.

Tolerance: float32 against float64 through up to 12 bounces: 2e-3 relative per sample (observed <= 2e-4; 1e-3 on one sample under
the environment's 60:1 hot texel).  A decision that hangs on
a float32-rounding-sized margin (a lobe pick, a roulette survival, a grazing hit) would change the whole remainder of that sample,
so a sample may differ outright — at most 1 % of them (observed: none of 900)."""
import copy
import os
import numpy as np
import pytest


def variants(scenes, vpt_scenes, vpt_mod):
    base = scenes("cornell_box")
    metal = copy.deepcopy(base)
    for k, m in enumerate(metal.materials):
        if not any(m["emissive_color"]):
            m.update(metallic=0.8 if k % 2 else 0.0, roughness=0.35 if k % 2 else 0.6, anisotropy=0.5 if k % 3 == 0 else 0.0)
    # an environment: 16 x 8 HDR texels with one hot "sun" texel and a dim gradient, rotated by azimuth / altitude (set by the test);
    # the light stays on, so both NEE strategies and both MIS weights are live
    sky = copy.deepcopy(base)
    rng = np.random.RandomState(3)
    env = np.zeros((8, 16, 4), np.float32)
    env[..., :3] = rng.gamma(0.8, 0.4, (8, 16, 3))
    env[2, 5, :3] = (60.0, 50.0, 40.0)
    sky.env = env
    # the glass sphere filled with a scattering medium (ClosestHit.slang:80-116): Henyey-Greenstein walk inside, stale pdf, no depth increment
    murky = copy.deepcopy(scenes("cornell_box_glass"))
    for m in murky.materials:
        if m["transmission"] > 0:
            m.update(roughness=0.2, medium_density=0.6, medium_anisotropy=0.4, medium_color=(0.9, 0.6, 0.3))
    # homogeneous box volumes (RayGen.slang:162-372): fog filling the room; two overlapping boxes of different density under the environment
    fog = [vpt_mod.volume(corner_min=(-5.0, -10.5, -5.0), corner_max=(5.0, -0.5, 5.0), color=(0.9, 0.9, 0.9), density=0.12, anisotropy=0.5)]
    two = [vpt_mod.volume(corner_min=(-5.0, -10.5, -5.0), corner_max=(1.0, -4.0, 5.0), color=(0.8, 0.9, 1.0), density=0.15, anisotropy=-0.3),
           vpt_mod.volume(corner_min=(-2.0, -7.0, -3.0), corner_max=(5.0, -0.5, 3.0), color=(1.0, 0.8, 0.7), density=0.3, anisotropy=0.0, emissive_color=(0.05, 0.02, 0.0))]
    # textures: the Viking room (a 1024 x 1024 base-colour map) and the textured boxes (base colour + a metallic-roughness map) under the
    # same environment
    viking = copy.deepcopy(scenes("viking_room")); viking.env = env
    boxes = copy.deepcopy(vpt_scenes.load_gltf(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "textured_boxes.gltf"))); boxes.env = env
    return {"cornell_d6": (base, 6, 90, 3), "cornell_d12": (base, 12, 60, 2), "metal_anisotropic": (metal, 6, 60, 2),
            "glass_sphere": (scenes("cornell_box_glass"), 8, 60, 2), "environment": (sky, 6, 90, 3),
            "textured_viking_room": (viking, 5, 60, 2), "textured_boxes": (boxes, 5, 60, 2), "medium_in_glass": (murky, 10, 80, 2),
            "fog": (base, 8, 80, 2, fog), "two_boxes_environment": (sky, 8, 80, 2, two), "depth_of_field_3spf": (base, 6, 80, 2),
            "atmosphere": (base, 10, 120, 2), "atmosphere_with_fog": (base, 10, 120, 2, fog),
            "fog_draine": (base, 8, 80, 2, fog), "fog_hg_plus_draine": (sky, 8, 80, 2, fog),
            "flags_no_mis_no_compensation": (sky, 6, 80, 2), "flags_geometry_normals_hidden_env": (sky, 6, 80, 2), "flags_furnace": (sky, 6, 80, 2),
            # SetUseRayQueries(false): the TraceRay forms of the shadow / distance queries (RTCommon.slang:64-84, 103-117) — under the environment
            # with the lamp on (its NEE sample is drawn and never visible), and with fog (the distance query)
            "flags_no_ray_queries": (sky, 6, 90, 3), "flags_no_ray_queries_fog": (sky, 8, 80, 2, fog),
            # heterogeneous boxes (Volume.slang:69-147, 299-348, 448-520): density from a grid — delta-tracked scattering block by block through
            # the 32^3 majorant table, ratio-tracked transmittance with roulette (it draws, so NEE order matters), jittered lookups; a smoke
            # column in the room, and the same column inside homogeneous fog under the environment
            "smoke_grid": (base, 8, 80, 2, [vpt_mod.volume(corner_min=(-3.0, -9.5, -3.0), corner_max=(3.0, -1.0, 3.0), color=(0.85, 0.85, 0.9), density=1.2, anisotropy=0.3, density_data_index=0)], "grid"),
            "smoke_grid_in_fog_environment": (sky, 8, 80, 2, [vpt_mod.volume(corner_min=(-5.0, -10.5, -5.0), corner_max=(5.0, -0.5, 5.0), color=(0.9, 0.9, 0.9), density=0.05, anisotropy=0.0),
                                                               vpt_mod.volume(corner_min=(-3.0, -9.5, -3.0), corner_max=(3.0, -1.0, 3.0), color=(0.9, 0.8, 0.7), density=0.9, anisotropy=-0.2, density_data_index=0, grid_sharpness=1.5)], "grid")}


def smoke_grid():
    """A 24 x 20 x 16 density grid (raw values up to 3.7, so the normalisation by the maximum is in play): a lumpy column with empty cells."""
    z, y, x = np.mgrid[0:16, 0:20, 0:24].astype(np.float64)
    r2 = ((x - 11.5) / 9.0) ** 2 + ((z - 7.5) / 6.0) ** 2
    d = np.clip(1.0 - r2, 0.0, None) * (0.4 + 0.6 * np.sin(y * 0.9) ** 2) * (1.0 + 0.5 * np.cos(x * 1.3 + z * 0.7))
    d[d < 0.08] = 0.0
    return (d * 3.7 / d.max()).astype(np.float32)


@pytest.mark.parametrize("which", ["cornell_d6", "cornell_d12", "metal_anisotropic", "glass_sphere", "environment", "textured_viking_room", "textured_boxes", "medium_in_glass", "fog", "two_boxes_environment", "depth_of_field_3spf", "atmosphere", "atmosphere_with_fog", "fog_draine", "fog_hg_plus_draine", "flags_no_mis_no_compensation", "flags_geometry_normals_hidden_env", "flags_furnace", "flags_no_ray_queries", "flags_no_ray_queries_fog", "smoke_grid", "smoke_grid_in_fog_environment"])
def test_per_sample_values_match_the_float64_integrator(vpt, oracle, scenes, which):
    import ref_integrator64 as R
    v = variants(scenes, vpt.scenes, vpt)[which]
    sc, depth, npix, frames = v[:4]
    vols = v[4] if len(v) > 4 else []
    grids = [smoke_grid()] if len(v) > 5 else []
    W, H = 64, 36
    P = vpt.default_params(max_depth=depth)
    if which in ("environment", "textured_viking_room", "textured_boxes", "two_boxes_environment", "fog_hg_plus_draine", "smoke_grid_in_fog_environment"):
        P = vpt.default_params(max_depth=depth, sky_azimuth=35.0, sky_altitude=-20.0, sky_intensity=1.5)
    luts = vpt.scenes.load_luts()
    if which == "depth_of_field_3spf":   # thin-lens offset on the camera plane, three samples per dispatch from one sampler
        P = vpt.default_params(max_depth=depth, dof_strength=0.6, focus_distance=14.0, samples_per_frame=3)
    if which.startswith("flags_"):   # the reference's feature #defines (vpt_params.flags)
        a = vpt._abi
        fl = {"flags_no_mis_no_compensation": a.FLAGS_DEFAULT & ~(a.FLAG_SKY_MIS | a.FLAG_MESH_MIS | a.FLAG_ENERGY_COMPENSATION),
              "flags_geometry_normals_hidden_env": (a.FLAGS_DEFAULT | a.FLAG_GEOMETRY_NORMALS) & ~a.FLAG_SHOW_ENV_DIRECTLY,
              "flags_furnace": a.FLAGS_DEFAULT | a.FLAG_FURNACE,
              "flags_no_ray_queries": a.FLAGS_DEFAULT & ~a.FLAG_RAY_QUERIES, "flags_no_ray_queries_fog": a.FLAGS_DEFAULT & ~a.FLAG_RAY_QUERIES}[which]
        P = vpt.default_params(max_depth=depth, sky_azimuth=35.0, sky_altitude=-20.0, sky_intensity=1.5, flags=fl)
    atm = None
    if which.startswith("atmosphere"):   # Rayleigh / Mie / ozone delta tracking, one colour channel per path after the first collision, sun-disk NEE
        P = vpt.default_params(max_depth=depth, sky_altitude=-55.0, sky_azimuth=160.0)
        atm = vpt.atmosphere()
    S = R.Scene64(sc, W, H); S.set_volumes(vols, grids); S.set_atmosphere(atm)
    S.phase = {"fog_draine": 1, "fog_hg_plus_draine": 2}.get(which, 0)
    o = oracle.Oracle(sc, W, H); o.set_params(P)
    for g in grids:
        assert o.add_density_grid(g) >= 0
    if vols:
        o.set_volumes(vols)
    if atm is not None:
        o.set_atmosphere(atm)
    if S.phase:
        o.set_phase_function(S.phase)
    rng = np.random.default_rng(4)
    lo_x, hi_x = (12, 52) if which in ("cornell_d6", "cornell_d12", "metal_anisotropic", "glass_sphere", "medium_in_glass", "fog", "depth_of_field_3spf", "smoke_grid") else (0, 64)                                        # with a sky, also the pixels beside the box
    xs = rng.integers(lo_x, hi_x, npix).astype(np.uint32); ys = rng.integers(4, 32, npix).astype(np.uint32)   # pixels that look into the box
    got = o.pixel_samples(xs, ys, 0, frames).astype(np.float64)
    o.close()
    bad, total, lit = 0, 0, 0
    lum_got, lum_ref = [], []
    for i, (x, y) in enumerate(zip(xs, ys)):
        for f in range(frames):
            ref = R.sample_value(S, luts, int(x), int(y), f, P)
            total += 1
            lit += bool(ref.max() > 0)
            lum_got.append(float(got[i, f].sum())); lum_ref.append(float(ref.sum()))
            # (atmosphere: exp(-height / falloff) of heights that float32 resolves to half a metre at a 6.36e6 m radius: 5e-3)
            if not np.allclose(got[i, f], ref, rtol=5e-3 if atm is not None else 2e-3, atol=1e-6):
                bad += 1
    assert lit > 0.5 * total            # the comparison is not about black pixels
    if not grids:
        assert bad <= 0.01 * total, (bad, total)
    else:
        # Density grids: the block walk re-enters the volume 1e-4 of the box beyond every block face, so a last-bit difference in the ray
        # (float64 here, float32 there) moves about one crossing in a thousand to the other side of a face — one random draw more or
        # less, and the rest of the sample is another sample (ref_integrator64.hetero_walk).  Per-sample equality therefore holds for
        # the samples without such an event (most of them: any difference in the LOGIC would break every sample that meets the grid),
        # and the two sets of samples must be draws of the same distribution: their means agree within the Monte-Carlo error.
        assert bad <= 0.25 * total, (bad, total)
        a, b = np.array(lum_got), np.array(lum_ref)
        se = np.sqrt((a.var() + b.var()) / len(a))
        assert abs(a.mean() - b.mean()) <= 4.0 * se + 1e-9, (a.mean(), b.mean(), se)
