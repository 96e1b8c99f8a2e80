"""Parity tests proper: the HIP backend, called through the C-ABI, against the CPU oracle on identical
scenes / seeds.  Bar (BASELINE.json north_star): <= 1e-4 relative L2 radiance error; because both sides
share the fp32 contract (include/vpt_fp32.h) the comparisons below demand BIT-EXACT images, which is
stronger; REL_L2_TOL is the stated tolerance a future change would be held to if exactness were lost."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL_L2_TOL = 1e-4


def rel_l2(a, b):
    return float(np.sqrt(((a.astype(np.float64) - b) ** 2).sum()) / max(np.sqrt((b.astype(np.float64) ** 2).sum()), 1e-30))


def render_both(vpt, oracle, sc, w, h, params, frames, **gpu_kw):
    o = oracle.Oracle(sc, w, h)
    o.set_params(params)
    o.render(frames)
    ref = o.radiance()
    ctr = o.counters()
    o.close()
    g = vpt.PathTracer(w, h, **gpu_kw)
    g.set_scene(sc)
    g.set_params(params)
    g.render(frames)
    img = g.radiance()
    st = g.stats()
    g.close()
    return img, ref, st, ctr


def assert_parity(img, ref):
    assert np.isfinite(img).all()
    assert rel_l2(img[..., :3], ref[..., :3]) <= REL_L2_TOL
    assert np.array_equal(img, ref), "bit-exact parity lost: %d px differ, relL2 %.3g" % (
        int((np.abs(img - ref).max(axis=2) > 0).sum()), rel_l2(img[..., :3], ref[..., :3]))


def config1_scene(vpt, scenes):
    """BASELINE config 1: Cornell box, 1 diffuse material (Khaki everywhere), constant white 64x32 env."""
    sc = copy.deepcopy(scenes("cornell_box"))
    for m in sc.materials:
        m.update(base_color=(0.8, 0.66, 0.44), emissive_color=(0, 0, 0), metallic=0.0, roughness=1.0, ior=1.5)
    sc.env = vpt.scenes.constant_env((1, 1, 1), 64, 32)
    return sc


def test_config1_cornell_256_16spp_depth4(vpt, oracle, scenes):
    img, ref, st, ctr = render_both(vpt, oracle, config1_scene(vpt, scenes), 256, 256, vpt.default_params(max_depth=4), 16)
    assert_parity(img, ref)
    assert st["closest_rays"] == ctr["closest"] and st["samples"] == 256 * 256 * 16


def test_cornell_emissive_depth8(vpt, oracle, scenes):
    """BASELINE config 2 at reduced size: Cornell as shipped (emissive 50), black env, depth 8."""
    img, ref, st, ctr = render_both(vpt, oracle, scenes("cornell_box"), 320, 180, vpt.default_params(max_depth=8), 8)
    assert_parity(img, ref)
    assert st["closest_rays"] == ctr["closest"]
    assert st["emissive_mesh_count"] == 1 and st["emissive_triangle_count"] == 2


def test_cornell_glass_depth32(vpt, oracle, scenes):
    """Dielectric sphere (transmission 1): refraction, medium enter/exit, total internal reflection."""
    img, ref, _, _ = render_both(vpt, oracle, scenes("cornell_box_glass"), 240, 135, vpt.default_params(max_depth=32), 6)
    assert_parity(img, ref)


def test_textured_scene_with_hdr_env(vpt, oracle, scenes):
    """VikingRoom: 1024^2 base-colour texture (bilinear REPEAT, gamma 2.2), env alias sampling + miss lookups."""
    sc = copy.deepcopy(scenes("viking_room"))
    sc.env = vpt.scenes.sun_sky_env(128, 64, seed=3, sun_peak=300.0)
    sc.view_inverse = sc.view_inverse.copy()
    sc.view_inverse[:3, 3] *= 0.5  # the scene's own camera, moved in so the model fills the frame
    img, ref, _, ctr = render_both(vpt, oracle, sc, 200, 120, vpt.default_params(max_depth=6, sky_azimuth=40.0, sky_altitude=-15.0, sky_intensity=1.5), 6)
    assert ctr["closest"] > 200 * 120 * 6 * 1.1 and ctr["shadow"] > 100000  # the camera actually sees the model
    assert_parity(img, ref)


@pytest.mark.parametrize("kw", [
    dict(samples_per_frame=3),                       # RNG stream carried across the samples of a pixel (RayGen.slang:33)
    dict(dof_strength=0.8, focus_distance=20.0),     # depth of field
    dict(max_luminance=2.0),                         # luminance clamp on every bounce but the first
    dict(emissive_pdf_bias=0.5),
    dict(max_depth=1),
    dict(max_depth=200),                             # reference default: Russian roulette is the only terminator
])
def test_parameter_variants(vpt, oracle, scenes, kw):
    p = vpt.default_params(max_depth=6)
    for k, v in kw.items():
        setattr(p, k, v)
    img, ref, _, _ = render_both(vpt, oracle, scenes("cornell_box"), 160, 90, p, 3)
    assert_parity(img, ref)


@pytest.mark.parametrize("clear,setf", [
    ("FLAG_SKY_MIS", None), ("FLAG_MESH_MIS", None), ("FLAG_SHOW_ENV_DIRECTLY", None), ("FLAG_ENERGY_COMPENSATION", None),
    (None, "FLAG_GEOMETRY_NORMALS"), (None, "FLAG_FURNACE"),
])
def test_feature_flags(vpt, oracle, scenes, clear, setf):
    """Each Slang #define of PathTracer.cpp:621-654 that touches the surface path."""
    sc = copy.deepcopy(scenes("cornell_box_glass"))
    sc.env = vpt.scenes.sun_sky_env(64, 32, seed=9, sun_peak=100.0)
    flags = vpt._abi.FLAGS_DEFAULT
    if clear:
        flags &= ~getattr(vpt._abi, clear)
    if setf:
        flags |= getattr(vpt._abi, setf)
    img, ref, _, _ = render_both(vpt, oracle, sc, 128, 72, vpt.default_params(max_depth=8, flags=flags), 3)
    assert_parity(img, ref)


def test_material_classes(vpt, oracle, scenes):
    """Metallic, anisotropic + rotated tangents, rough glass with a scattering medium, emissive: one wall each."""
    sc = copy.deepcopy(scenes("cornell_box"))
    sc.materials[0].update(metallic=1.0, roughness=0.3, anisotropy=0.7, anisotropy_rotation=35.0)
    sc.materials[1].update(transmission=1.0, roughness=0.2, ior=1.33, medium_density=0.4, medium_anisotropy=0.3, medium_color=(0.9, 0.5, 0.4))
    sc.materials[2].update(metallic=0.5, roughness=0.6, specular_color=(0.9, 0.8, 0.7))
    sc.env = vpt.scenes.sun_sky_env(64, 32, seed=2, sun_peak=50.0)
    img, ref, _, _ = render_both(vpt, oracle, sc, 160, 90, vpt.default_params(max_depth=12), 4)
    assert_parity(img, ref)


@pytest.mark.parametrize("name", ["cornell_box", "cornell_box_glass", "viking_room"])
@pytest.mark.parametrize("pipeline", [1, 2])
def test_both_pipelines_match_the_oracle(vpt, oracle, scenes, name, pipeline):
    """Bounces >= 1 run either fused (one kernel per bounce) or staged (extend / shade / connect with compacted
    queues); AUTO picks by BVH size, so both are forced here on small and large scenes alike."""
    sc = copy.deepcopy(scenes(name))
    sc.env = vpt.scenes.sun_sky_env(64, 32, seed=4, sun_peak=80.0)
    p = vpt.default_params(max_depth=10, samples_per_frame=2)
    img, ref, st, ctr = render_both(vpt, oracle, sc, 144, 81, p, 3, pipeline=pipeline, build_flags=4 if pipeline == 2 else 0)
    assert_parity(img, ref)
    assert st["closest_rays"] == ctr["closest"]
    launched = st["kernel_launches"]
    assert (launched["bounce"] > 0) == (pipeline == 1) and (launched["extend"] > 0) == (pipeline == 2)


@pytest.mark.parametrize("S,w,h", [(2, 64, 36), (3, 50, 31)])
@pytest.mark.parametrize("pipeline", [1, 2])
def test_split_screen_dispatch(vpt, oracle, scenes, S, w, h, pipeline):
    """SetSplitScreenCount (RayGen.slang:16-25, PathTracer.cpp:141-153): S^2 interleaved dispatches per frame, the first
    one filling whole S x S cells; the image must match after EVERY dispatch count, mid-frame included."""
    sc = scenes("cornell_box")
    p = vpt.default_params(max_depth=4, screen_chunk_count=S, samples_per_frame=2)
    o = oracle.Oracle(sc, w, h); o.set_params(p)
    g = vpt.PathTracer(w, h, pipeline=pipeline, frames_in_flight=5, build_flags=4 if pipeline == 2 else 0); g.set_scene(sc); g.set_params(p)
    for n in (1, 2, S * S - 3 if S > 2 else 1, S * S + 2, 7):
        o.render(n); g.render(n)
        assert np.array_equal(g.radiance(), o.radiance()), n
    st = g.stats()
    total = 1 + 2 + (S * S - 3 if S > 2 else 1) + S * S + 2 + 7
    assert st["dispatches"] == total and st["frames"] == total // (S * S)
    o.close(); g.close()


def test_frames_in_flight_do_not_change_the_image(vpt, oracle, scenes):
    """Several frames share one wavefront batch; the running mean is still applied in frame order."""
    sc = scenes("cornell_box")
    p = vpt.default_params(max_depth=5)
    imgs = []
    for fif in (1, 3, 8):
        g = vpt.PathTracer(96, 54, frames_in_flight=fif)
        g.set_scene(sc); g.set_params(p)
        g.render(5); g.render(3)   # 8 frames over two calls
        imgs.append(g.radiance())
        assert g.stats()["frames"] == 8
        g.close()
    assert np.array_equal(imgs[0], imgs[1]) and np.array_equal(imgs[0], imgs[2])
    o = oracle.Oracle(sc, 96, 54); o.set_params(p); o.render(8)
    assert np.array_equal(imgs[0], o.radiance())
    o.close()


def test_row_shards_reassemble_to_the_unsharded_image(vpt, scenes):
    """Multi-GPU partition (rows y % G == g): seeds depend on pixel and frame only, so any G is bit-identical."""
    import ctypes as C
    sc = scenes("cornell_box")
    p = vpt.default_params(max_depth=5)
    W, H = 100, 37  # ragged: 37 rows over 3 shards
    g = vpt.PathTracer(W, H); g.set_scene(sc); g.set_params(p); g.render(3)
    whole = g.radiance(); g.close()
    G = 3
    parts = []
    for r in range(G):
        s = vpt.PathTracer(W, H, shard_rank=r, shard_count=G)
        s.set_scene(sc); s.set_params(p); s.render(3)
        parts.append(s)
    n = parts[0].shard_floats()
    assert n == ((H + G - 1) // G) * W * 4
    lib = vpt.load_library()
    hip = C.CDLL("libamdhip64.so")
    buf = C.c_void_p()
    assert hip.hipMalloc(C.byref(buf), n * 4 * G) == 0
    for r, s in enumerate(parts):
        s.shard_to_device(C.c_void_p(buf.value + r * n * 4))
    parts[0].assemble_shards(buf, G)
    assert np.array_equal(parts[0].radiance(), whole)
    hip.hipFree(buf)
    for s in parts:
        s.close()


def test_set_material_and_reset_semantics(vpt, oracle, scenes):
    """SetMaterial (PathTracer.cpp:712-810): emissive list rebuilt, accumulation reset; max_samples stops rendering."""
    sc = scenes("cornell_box")
    g = vpt.PathTracer(64, 36); g.set_scene(sc)
    p = vpt.default_params(max_depth=4, max_samples=3)
    g.set_params(p)
    assert g.render(2) is False
    assert g.render(5) is True           # PathTrace returns true once max samples are reached...
    assert g.stats()["frames"] == 3      # ...and launches nothing more
    m = g.get_material(2)
    m.emissive_color[:] = [3.0, 2.0, 1.0]
    g.set_material(2, m)
    st = g.stats()
    assert st["frames"] == 0 and st["emissive_mesh_count"] == 4 and st["emissive_triangle_count"] == 8
    g.render(2)
    img = g.radiance()
    sc2 = copy.deepcopy(sc); sc2.materials[2]["emissive_color"] = (3.0, 2.0, 1.0)
    o = oracle.Oracle(sc2, 64, 36); o.set_params(p); o.render(2)
    assert np.array_equal(img, o.radiance())
    o.close(); g.close()


def test_checkpoint_resume(vpt, scenes):
    """Accumulation image + frame counter are the whole progressive state."""
    sc = scenes("cornell_box")
    p = vpt.default_params(max_depth=4)
    a = vpt.PathTracer(64, 36); a.set_scene(sc); a.set_params(p); a.render(6)
    full = a.radiance()
    b = vpt.PathTracer(64, 36); b.set_scene(sc); b.set_params(p); b.render(4)
    snap = b.radiance(); b.close()
    c = vpt.PathTracer(64, 36); c.set_scene(sc); c.set_params(p)
    c.set_radiance(snap, 4); c.render(2)
    assert np.array_equal(c.radiance(), full)
    a.close(); c.close()


def test_errors(vpt, scenes):
    g = vpt.PathTracer(32, 32)
    with pytest.raises(vpt.VptError, match="NO_SCENE"):
        g.render(1)
    with pytest.raises(vpt.VptError, match="INVALID_ARGUMENT"):
        g.set_params(vpt.default_params(screen_chunk_count=0))
    with pytest.raises(vpt.VptError, match="INVALID_ARGUMENT"):
        g.set_params(vpt.default_params(max_depth=0))
    s2 = vpt.PathTracer(32, 32, shard_rank=0, shard_count=2)
    with pytest.raises(vpt.VptError, match="UNSUPPORTED"):
        s2.set_params(vpt.default_params(screen_chunk_count=2))  # split-screen copies pixels across rows: one context only
    s2.close()
    bad = copy.deepcopy(scenes("cornell_box"))
    bad.materials[0]["base_color_texture"] = 99
    with pytest.raises(vpt.VptError, match="INVALID_ARGUMENT"):
        g.set_scene(bad)
    g.close()
