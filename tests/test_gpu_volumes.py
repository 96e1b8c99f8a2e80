"""Homogeneous box volumes (RayGen.slang:162-380, Volume.slang) — HIP fused pipeline vs the oracle,
bit-exact, on the Cornell box with fog / smoke boxes, all three phase functions, both NEE flags."""
import copy

import numpy as np
import pytest

from test_gpu_parity import assert_parity

pytestmark = pytest.mark.gpu


def render_both(vpt, oracle, sc, w, h, params, frames, volumes, phase=0, **gpu_kw):
    o = oracle.Oracle(sc, w, h)
    o.set_params(params); o.set_volumes(volumes); o.set_phase_function(phase)
    o.render(frames)
    ref = o.radiance(); o.close()
    g = vpt.PathTracer(w, h, **gpu_kw)
    g.set_scene(sc); g.set_params(params); g.set_volumes(volumes); g.set_phase_function(phase)
    g.render(frames)
    img = g.radiance(); st = g.stats(); g.close()
    return img, ref, st


def fog(vpt, **kw):
    """A box a little smaller than the Cornell room (walls at +-5.68, floor..ceiling 0..-11.4 in the Y-down world)."""
    d = dict(corner_min=(-5.0, -10.5, -5.0), corner_max=(5.0, -0.5, 5.0), color=(0.9, 0.85, 0.8), density=0.12, anisotropy=0.3)
    d.update(kw)
    return vpt.volume(**d)


def lit_env_scene(vpt, scenes):
    sc = copy.deepcopy(scenes("cornell_box"))
    sc.env = vpt.scenes.sun_sky_env(64, 32, seed=5, sun_peak=200.0)
    return sc


@pytest.mark.parametrize("phase", [0, 1, 2])
def test_fog_in_cornell_all_phase_functions(vpt, oracle, scenes, phase):
    vols = [fog(vpt, alpha=0.6, droplet_size=12.0)]
    img, ref, st = render_both(vpt, oracle, scenes("cornell_box"), 192, 108, vpt.default_params(max_depth=8), 4, vols, phase)
    assert_parity(img, ref)
    assert st["kernel_launches"]["bounce"] > 0 and st["kernel_launches"]["extend"] == 0  # volumes run fused


def test_volume_changes_the_image_and_removal_restores_it(vpt, oracle, scenes):
    sc = scenes("cornell_box")
    P = vpt.default_params(max_depth=6)
    g = vpt.PathTracer(160, 90); g.set_scene(sc); g.set_params(P)
    g.render(3); base = g.radiance()
    g.set_volumes([fog(vpt)]); g.render(3); fogged = g.radiance()
    g.set_volumes([]); g.render(3); again = g.radiance(); g.close()
    assert np.array_equal(base, again) and not np.array_equal(base, fogged)
    o = oracle.Oracle(sc, 160, 90); o.set_params(P); o.render(3)
    assert np.array_equal(base, o.radiance()); o.close()


def test_overlapping_emissive_and_dense_volumes_with_env(vpt, oracle, scenes):
    """Three boxes: room fog, a dense smoke column overlapping it (ties in the entry-distance sort), an emissive
    box outside the room seen through the open front; sun-and-sky env so sky NEE crosses volumes."""
    sc = lit_env_scene(vpt, scenes)
    vols = [fog(vpt),
            vpt.volume(corner_min=(-1.5, -9.0, -1.0), corner_max=(1.0, -0.2, 1.5), color=(0.3, 0.3, 0.35), density=1.5, anisotropy=-0.4),
            vpt.volume(corner_min=(-8.0, -6.0, 7.0), corner_max=(-6.0, -4.0, 9.0), color=(0.5, 0.5, 0.5), emissive_color=(2.0, 1.0, 0.3), density=0.8,
                       approximated_scattering=1, anisotropy=0.7)]
    img, ref, _ = render_both(vpt, oracle, sc, 192, 108, vpt.default_params(max_depth=12), 4, vols)
    assert_parity(img, ref)


@pytest.mark.parametrize("flags_off", ["VPT_FLAG_SKY_MIS", "VPT_FLAG_MESH_MIS", "both"])
def test_volume_nee_flags(vpt, oracle, scenes, flags_off):
    from importlib import import_module
    abi = import_module("vulkan-path-tracer_amd._abi")
    off = {"VPT_FLAG_SKY_MIS": abi.FLAG_SKY_MIS, "VPT_FLAG_MESH_MIS": abi.FLAG_MESH_MIS, "both": abi.FLAG_SKY_MIS | abi.FLAG_MESH_MIS}[flags_off]
    P = vpt.default_params(max_depth=8)
    P.flags &= ~off
    img, ref, _ = render_both(vpt, oracle, lit_env_scene(vpt, scenes), 160, 90, P, 3, [fog(vpt, density=0.25)])
    assert_parity(img, ref)


def test_camera_inside_volume_glass_scene_multisample_and_shards(vpt, oracle, scenes):
    """Camera inside a large thin fog (entry distance 0), glass sphere (surface media + box volumes together),
    4 samples per frame (VolumeDepth resets per sample), and a 2-way row shard assembled on the host."""
    sc = scenes("cornell_box_glass")
    vols = [vpt.volume(corner_min=(-30, -30, -30), corner_max=(30, 30, 30), color=(0.95, 0.95, 1.0), density=0.02, anisotropy=0.8)]
    P = vpt.default_params(max_depth=16, samples_per_frame=4)
    img, ref, _ = render_both(vpt, oracle, sc, 128, 72, P, 2, vols)
    assert_parity(img, ref)
    import ctypes as C
    parts = []
    for r in range(2):
        g = vpt.PathTracer(128, 72, shard_rank=r, shard_count=2)
        g.set_scene(sc); g.set_params(P); g.set_volumes(vols); g.render(2)
        parts.append(g)
    n = parts[0].shard_floats()
    hip = C.CDLL("libamdhip64.so")
    buf = C.c_void_p()
    assert hip.hipMalloc(C.byref(buf), n * 4 * 2) == 0
    for r, g in enumerate(parts):
        g.shard_to_device(C.c_void_p(buf.value + r * n * 4))
    parts[0].assemble_shards(buf, 2)
    assert np.array_equal(parts[0].radiance(), ref)
    hip.hipFree(buf)
    for g in parts:
        g.close()


def test_volume_argument_errors(vpt, scenes):
    g = vpt.PathTracer(32, 18); g.set_scene(scenes("cornell_box"))
    with pytest.raises(vpt.VptError, match="INVALID"):
        v = vpt.volume(); v.density_data_index = 0   # no grid was added
        g.set_volumes([v])
    with pytest.raises(vpt.VptError, match="LIMIT"):
        g.set_volumes([vpt.volume()] * 33)
    with pytest.raises(vpt.VptError, match="INVALID"):
        g.set_volumes([vpt.volume(density=0.0)])
    with pytest.raises(vpt.VptError, match="INVALID"):
        g.set_phase_function(3)
    g.close()
    s = vpt.PathTracer(32, 18, pipeline=2); s.set_scene(scenes("cornell_box"))   # a BVH that rides in LDS has no streams pipeline
    with pytest.raises(vpt.VptError, match="UNSUPPORTED"):
        s.set_volumes([vpt.volume()])
    s.close()
    for pipe in (3, 4) if vpt.has_lab() else (4,):   # round 1's stage kernels (laboratory build) and the class sort never run media
        s = vpt.PathTracer(32, 18, pipeline=pipe); s.set_scene(scenes("cornell_box_glass"))
        with pytest.raises(vpt.VptError, match="UNSUPPORTED"):
            s.set_volumes([vpt.volume()])
        s.close()


def glass_room(vpt, scenes, lit=True):
    """The Cornell room with the 960-triangle glass sphere: its BVH lives in memory, so AUTO runs media on the streams."""
    sc = copy.deepcopy(scenes("cornell_box_glass"))
    if lit:
        sc.env = vpt.scenes.sun_sky_env(64, 32, seed=5, sun_peak=200.0)
    return sc


@pytest.mark.parametrize("phase", [0, 1, 2])
def test_media_on_the_streams_pipeline_fog_all_phase_functions(vpt, oracle, scenes, phase):
    """kernels_media.hip: distance -> scatter -> extend -> shade -> shadow rays -> tail on the stream pipeline's queues, forced
    (pipeline 2) and by AUTO, against the oracle AND against the fused media kernel (pipeline 1) on the same scene: fog box +
    dense smoke column + glass sphere (surface medium and box volumes together) under a lit environment."""
    sc = glass_room(vpt, scenes)
    vols = [fog(vpt, alpha=0.6, droplet_size=12.0),
            vpt.volume(corner_min=(-1.5, -9.0, -1.0), corner_max=(1.0, -0.2, 1.5), color=(0.3, 0.3, 0.35), density=1.5, anisotropy=-0.4)]
    P = vpt.default_params(max_depth=10)
    img, ref, st = render_both(vpt, oracle, sc, 160, 90, P, 3, vols, phase, pipeline=2)
    assert_parity(img, ref)
    assert st["kernel_launches"]["bounce"] == 0 and st["kernel_launches"]["extend"] > 0 and st["kernel_launches"]["join"] > 0   # the media stages ran
    for pipe in (0, 1):
        g = vpt.PathTracer(160, 90, pipeline=pipe); g.set_scene(sc); g.set_params(P); g.set_volumes(vols); g.set_phase_function(phase); g.render(3)
        assert np.array_equal(g.radiance(), ref); st2 = g.stats(); g.close()
        assert (st2["kernel_launches"]["bounce"] > 0) == (pipe == 1)   # AUTO picks the streams for a BVH in memory, 1 is the fused kernel


def test_media_on_the_streams_multisample_nee_flags_and_emissive_box(vpt, oracle, scenes):
    """4 samples per frame (next-sample regeneration in the tail stage), a camera inside a thin fog, an emissive box, each NEE
    flag off in turn (the light-identity shadow ray that also counts a clean miss needs the mesh sample), 129 x 73 (ragged tiles)."""
    from importlib import import_module
    abi = import_module("vulkan-path-tracer_amd._abi")
    sc = glass_room(vpt, scenes)
    vols = [vpt.volume(corner_min=(-30, -30, -30), corner_max=(30, 30, 30), color=(0.95, 0.95, 1.0), density=0.02, anisotropy=0.8),
            vpt.volume(corner_min=(-3.0, -6.0, 1.0), corner_max=(-1.0, -4.0, 3.0), color=(0.5, 0.5, 0.5), emissive_color=(2.0, 1.0, 0.3), density=0.8,
                       approximated_scattering=1, anisotropy=0.7)]
    for off in (0, abi.FLAG_SKY_MIS, abi.FLAG_MESH_MIS):
        P = vpt.default_params(max_depth=12, samples_per_frame=4)
        P.flags &= ~off
        img, ref, _ = render_both(vpt, oracle, sc, 129, 73, P, 2, vols, pipeline=2)
        assert_parity(img, ref)


def test_media_on_the_streams_heterogeneous_and_atmosphere(vpt, oracle, scenes):
    """Everything that TRACKS its transmittance (random draws after the visibility bits are known): a grid cloud, the atmosphere's
    sun NEE and collisions with the colour-channel split, the room with the glass sphere — on the streams."""
    from test_oracle_volumes import cloud_grid
    sc = glass_room(vpt, scenes, lit=False)
    P = vpt.default_params(max_depth=8, sky_altitude=-50.0, sky_azimuth=150.0, samples_per_frame=2)
    o = oracle.Oracle(sc, 128, 72); o.set_params(P); gi = o.add_density_grid(cloud_grid(seed=5))
    g = vpt.PathTracer(128, 72, pipeline=2); g.set_scene(sc); g.set_params(P); g.add_density_grid(cloud_grid(seed=5))
    vols = [vpt.volume(corner_min=(-4.0, -9.0, -4.0), corner_max=(4.0, -2.0, 4.0), color=(0.9, 0.9, 0.9), density=1.0, density_data_index=gi),
            fog(vpt, density=0.05)]
    for x in (o, g):
        x.set_volumes(vols); x.set_atmosphere(vpt.atmosphere()); x.render(3)
    ref = o.radiance(); img = g.radiance(); st = g.stats(); o.close(); g.close()
    assert_parity(img, ref)
    assert st["kernel_launches"]["join"] > 0 and st["kernel_launches"]["bounce"] == 0


@pytest.mark.parametrize("approx", [0, 1])
def test_heterogeneous_cloud_in_cornell(vpt, oracle, scenes, approx):
    """A dense-grid smoke cloud (delta-tracked scattering, ratio-tracked NEE transmittance — both draw random numbers,
    the latter only for unobscured samples) next to a homogeneous fog box, under a lit env."""
    from test_oracle_volumes import cloud_grid
    sc = lit_env_scene(vpt, scenes)
    grid = cloud_grid()
    P = vpt.default_params(max_depth=10)
    o = oracle.Oracle(sc, 160, 90); o.set_params(P)
    gi = o.add_density_grid(grid)
    g = vpt.PathTracer(160, 90); g.set_scene(sc); g.set_params(P)
    assert g.add_density_grid(grid) == gi == 0
    vols = [vpt.volume(corner_min=(-3.5, -8.0, -3.0), corner_max=(3.0, -1.0, 3.5), color=(0.85, 0.85, 0.9), density=1.6, anisotropy=0.5,
                       density_data_index=gi, grid_sharpness=1.3, approximated_scattering=approx, approximated_scattering_falloff=0.7),
            fog(vpt, density=0.05)]
    o.set_volumes(vols); g.set_volumes(vols)
    o.render(4); g.render(4)
    ref = o.radiance(); img = g.radiance(); o.close()
    assert_parity(img, ref)
    with pytest.raises(vpt.VptError, match="INVALID"):
        g.clear_density_grids()                      # still referenced
    g.set_volumes([fog(vpt)]); g.clear_density_grids()
    with pytest.raises(vpt.VptError, match="INVALID"):
        g.set_volumes([vpt.volume(density_data_index=0)])   # no such grid any more
    g.close()


def test_heterogeneous_cloud_under_the_atmosphere(vpt, oracle, scenes):
    """Everything that tracks at once: atmosphere collisions + sun NEE, a grid cloud, the Cornell room."""
    from test_oracle_volumes import cloud_grid
    sc = scenes("cornell_box")
    P = vpt.default_params(max_depth=8, sky_altitude=-50.0, sky_azimuth=150.0)
    o = oracle.Oracle(sc, 128, 72); o.set_params(P); gi = o.add_density_grid(cloud_grid(seed=5))
    g = vpt.PathTracer(128, 72); g.set_scene(sc); g.set_params(P); g.add_density_grid(cloud_grid(seed=5))
    vols = [vpt.volume(corner_min=(-4.0, -9.0, -4.0), corner_max=(4.0, -2.0, 4.0), color=(0.9, 0.9, 0.9), density=1.0, density_data_index=gi)]
    for x in (o, g):
        x.set_volumes(vols); x.set_atmosphere(vpt.atmosphere()); x.render(3)
    ref = o.radiance(); img = g.radiance(); o.close(); g.close()
    assert_parity(img, ref)


@pytest.mark.parametrize("blackbody", [1, 0])
def test_fire_volume_temperature_emission(vpt, oracle, scenes, blackbody):
    """Emission from temperature (Volume.slang:233-258): blackbody colour ramp or a fixed colour, intensity
    temperature^gamma * scale, read from the volume's (merged) grid with the jittered lookup."""
    from test_oracle_volumes import cloud_grid
    sc = scenes("cornell_box")
    P = vpt.default_params(max_depth=6)
    o = oracle.Oracle(sc, 128, 72); o.set_params(P); gi = o.add_density_grid(cloud_grid(seed=9))
    g = vpt.PathTracer(128, 72); g.set_scene(sc); g.set_params(P); g.add_density_grid(cloud_grid(seed=9))
    fire = vpt.volume(corner_min=(-3.0, -7.0, -3.0), corner_max=(3.0, -0.5, 3.0), color=(0.2, 0.2, 0.2), density=1.2, density_data_index=gi,
                      has_temperature_data=1, use_blackbody=blackbody, temperature_color=(1.0, 0.4, 0.1), temperature_gamma=1.7, temperature_scale=6.0,
                      emissive_color_gamma=2.2, kelvin_min=800, kelvin_max=7000)
    for x in (o, g):
        x.set_volumes([fire]); x.render(3)
    ref = o.radiance(); img = g.radiance()
    assert_parity(img, ref)
    cold = vpt.volume(corner_min=(-3.0, -7.0, -3.0), corner_max=(3.0, -0.5, 3.0), color=(0.2, 0.2, 0.2), density=1.2, density_data_index=gi)
    g.set_volumes([cold]); g.render(3)
    assert g.radiance()[..., :3].sum() < img[..., :3].sum()      # the fire adds light
    with pytest.raises(vpt.VptError, match="INVALID"):
        g.set_volumes([vpt.volume(has_temperature_data=1)])     # temperature without a grid
    o.close(); g.close()
