"""VPT_PIPELINE_WHOLE (kernels_path.hip k_whole): a batch as ONE launch — persistent waves run every path from its camera ray to its
end, a lane whose path has ended takes the batch's next sample.  It is the reference's own shape (one RayGen thread = one whole
path, RayGen.slang:66-114) and must give the oracle's image bit for bit, with the per-bounce pipeline's ray statistics."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def oracle_image(oracle, sc, w, h, params, frames):
    o = oracle.Oracle(sc, w, h)
    o.set_params(params)
    o.render(frames)
    ref = o.radiance()
    ctr = o.counters()
    o.close()
    return ref, ctr


def gpu_image(vpt, sc, w, h, params, batches, **kw):
    g = vpt.PathTracer(w, h, **kw)
    g.set_scene(sc); g.set_params(params)
    for n in batches:
        g.render(n)
    img, st = g.radiance(), g.stats()
    g.close()
    return img, st


def scene_of(vpt, scenes, name):
    """cornell_box as shipped (every texture 1x1, black environment: the PLAIN instantiation), or the same 12 triangles — the golden
    Cornell box with the glass sphere keeps its BVH in memory, where the whole-path launch does not apply — with a glass wall, a rough
    glass floor holding a scattering medium and a sky (the general instantiation)."""
    if name == "cornell_box":
        return scenes(name)
    sc = copy.deepcopy(scenes("cornell_box"))
    sc.materials[0].update(transmission=1.0, roughness=0.05, ior=1.5, base_color=(1, 1, 1))
    sc.materials[2].update(transmission=1.0, roughness=0.3, ior=1.33, medium_density=0.6, medium_anisotropy=0.3, medium_color=(0.9, 0.5, 0.4))
    sc.env = vpt.scenes.sun_sky_env(64, 32, seed=9, sun_peak=100.0)
    return sc


RAY_STATS = ("closest_rays", "shadow_rays", "primary_hits", "primary_survivors", "primary_shadow_rays", "samples", "frames")


@pytest.mark.parametrize("name,depth,w,h", [("cornell_box", 8, 160, 90), ("cornell_box", 1, 97, 53), ("cornell_box", 200, 96, 54),
                                            ("cornell_glass", 12, 128, 72), ("cornell_glass", 40, 96, 54), ("cornell_box", 8, 8, 8)])
def test_whole_path_launches_equal_the_oracle_and_the_per_bounce_pipeline(vpt, oracle, scenes, name, depth, w, h):
    """The Cornell box runs the scene-class instantiation (PLAIN), the glass variant the general one (refraction, the medium state of
    a slot, environment lookups); 8 x 8 is less than one wave's first 64 samples, 97 x 53 leaves ragged tails everywhere."""
    sc = scene_of(vpt, scenes, name)
    p = vpt.default_params(max_depth=depth)
    ref, ctr = oracle_image(oracle, sc, w, h, p, 7)
    img, st = gpu_image(vpt, sc, w, h, p, [3, 3, 1], pipeline=vpt._abi.PIPELINE_WHOLE, frames_in_flight=3)
    assert np.array_equal(img, ref), "%d px differ" % int((np.abs(img - ref).max(axis=2) > 0).sum())
    assert st["closest_rays"] == ctr["closest"]
    fused, sf = gpu_image(vpt, sc, w, h, p, [3, 3, 1], pipeline=vpt._abi.PIPELINE_FUSED, frames_in_flight=3)
    assert np.array_equal(img, fused)
    for k in RAY_STATS:
        assert st[k] == sf[k], (k, st[k], sf[k])
    assert st["kernel_launches"]["primary"] == 3 and st["kernel_launches"]["bounce"] == 0 and st["kernel_launches"]["resolve"] == 3
    assert sf["kernel_launches"]["bounce"] > 0


def test_whole_path_general_kernel_with_textures_environment_and_every_material_class(vpt, oracle, scenes):
    """Metallic / anisotropic / rough glass with a scattering medium (in-medium events leave the depth alone: the batch has no fixed
    number of bounces, which a whole-path launch does not care about) / emissive, a sun-and-sky environment: the general instantiation."""
    sc = copy.deepcopy(scenes("cornell_box"))
    sc.materials[0].update(metallic=1.0, roughness=0.3, anisotropy=0.7, anisotropy_rotation=35.0)
    sc.materials[1].update(transmission=1.0, roughness=0.2, ior=1.33, medium_density=0.4, medium_anisotropy=0.3, medium_color=(0.9, 0.5, 0.4))
    sc.materials[2].update(metallic=0.5, roughness=0.6, specular_color=(0.9, 0.8, 0.7))
    sc.env = vpt.scenes.sun_sky_env(64, 32, seed=2, sun_peak=50.0)
    p = vpt.default_params(max_depth=12, sky_azimuth=30.0, sky_altitude=-10.0)
    ref, _ = oracle_image(oracle, sc, 160, 90, p, 4)
    img, st = gpu_image(vpt, sc, 160, 90, p, [4], pipeline=vpt._abi.PIPELINE_WHOLE)
    assert np.array_equal(img, ref)
    assert st["kernel_launches"]["primary"] == 1 and st["kernel_launches"]["bounce"] == 0


@pytest.mark.parametrize("kw", [dict(dof_strength=0.8, focus_distance=20.0), dict(max_luminance=2.0), dict(emissive_pdf_bias=0.5), dict(screen_chunk_count=2),
                                dict(flags="local_hits"), dict(flags="no_mis"), dict(flags="furnace")])
def test_whole_path_parameter_and_flag_variants(vpt, oracle, scenes, kw):
    sc = scene_of(vpt, scenes, "cornell_glass")
    p = vpt.default_params(max_depth=6)
    for k, v in kw.items():
        if k == "flags":
            A = vpt._abi
            p.flags = {"local_hits": A.FLAGS_DEFAULT | A.FLAG_LOCAL_HITS, "no_mis": A.FLAGS_DEFAULT & ~A.FLAG_SKY_MIS & ~A.FLAG_MESH_MIS,
                       "furnace": A.FLAGS_DEFAULT | A.FLAG_FURNACE}[v]
        else:
            setattr(p, k, v)
    frames = 8 if "screen_chunk_count" in kw else 3      # split-screen: a dispatch covers a quarter of the pixels
    ref, _ = oracle_image(oracle, sc, 128, 72, p, frames)
    img, _ = gpu_image(vpt, sc, 128, 72, p, [frames], pipeline=vpt._abi.PIPELINE_WHOLE)
    assert np.array_equal(img, ref)


def test_whole_path_counting_instantiations_count_what_the_per_bounce_kernels_count(vpt, scenes):
    """vpt_config.count_traversal: node and triangle visits are per ray, so the totals cannot depend on the schedule."""
    sc, p = scene_of(vpt, scenes, "cornell_glass"), vpt.default_params(max_depth=10)
    for flags in (vpt._abi.FLAGS_DEFAULT, vpt._abi.FLAGS_DEFAULT | vpt._abi.FLAG_LOCAL_HITS):
        p.flags = flags
        a, sa = gpu_image(vpt, sc, 96, 54, p, [4], pipeline=vpt._abi.PIPELINE_WHOLE, count_traversal=True)
        b, sb = gpu_image(vpt, sc, 96, 54, p, [4], pipeline=vpt._abi.PIPELINE_FUSED, count_traversal=True)
        assert np.array_equal(a, b)
        for k in ("nodes_visited", "tris_tested", "shadow_nodes_visited", "shadow_tris_tested") + RAY_STATS:
            assert sa[k] == sb[k] and sa[k] > 0, (k, sa[k], sb[k])


def test_auto_takes_the_whole_path_launch_wherever_it_applies(vpt, oracle, scenes):
    """AUTO: every batch of an LDS-resident scene without media is one launch (measured faster at every batch size,
    profiles/r04_whole_ab.json); VPT_LAB_WHOLE_FRAMES bounds the batch size it is taken for (0: the per-bounce kernels — the A/B switch).
    Same image whichever way the frames were grouped and run."""
    sc, w, h = scenes("cornell_box"), 128, 72
    p = vpt.default_params(max_depth=8)
    ref, _ = oracle_image(oracle, sc, w, h, p, 7)
    g = vpt.PathTracer(w, h, frames_in_flight=4)
    g.set_scene(sc); g.set_params(p)
    g.render(1); g.render(3)
    st = g.stats()
    assert st["kernel_launches"]["primary"] == 2 and st["kernel_launches"]["bounce"] == 0
    if not vpt.has_lab():   # the A/B switch below is a laboratory entry point (include/vpt_lab.h)
        g.render(3)
        assert g.stats()["kernel_launches"]["bounce"] == 0 and np.array_equal(g.radiance(), ref)
        g.close()
        return
    g.lab_set(vpt._abi.LAB_WHOLE_FRAMES, 1)
    g.render(2)
    st = g.stats()
    assert st["kernel_launches"]["primary"] == 3 and st["kernel_launches"]["bounce"] == 7
    g.lab_set(vpt._abi.LAB_WHOLE_FRAMES, 0)
    g.render(1)
    assert g.stats()["kernel_launches"]["bounce"] == 14
    g.lab_set(vpt._abi.LAB_WHOLE_FRAMES, 0xffff)
    g.render(4)
    st = g.stats()
    assert st["kernel_launches"]["bounce"] == 14 and st["kernel_launches"]["primary"] == 5 and st["frames"] == 11
    g.close()
    g = vpt.PathTracer(w, h, frames_in_flight=4)
    g.set_scene(sc); g.set_params(p)
    g.render(1); g.render(3)
    g.lab_set(vpt._abi.LAB_WHOLE_FRAMES, 1)
    g.render(2)
    g.lab_set(vpt._abi.LAB_WHOLE_FRAMES, 0)
    g.render(1)
    assert np.array_equal(g.radiance(), ref)
    g.close()


def test_whole_path_where_it_does_not_apply(vpt, scenes):
    """Asked for explicitly it fails loudly (no silent other pipeline); AUTO simply does not take it."""
    sc = scenes("cornell_box_glass")                   # the 960-triangle glass sphere: BVH in memory
    g = vpt.PathTracer(64, 36, pipeline=vpt._abi.PIPELINE_WHOLE)
    g.set_scene(sc); g.set_params(vpt.default_params(max_depth=4))
    with pytest.raises(vpt.VptError, match="VPT_PIPELINE_WHOLE"):
        g.render(1)
    g.close()
    g = vpt.PathTracer(64, 36, pipeline=vpt._abi.PIPELINE_WHOLE)
    g.set_scene(scenes("cornell_box")); g.set_params(vpt.default_params(max_depth=4, samples_per_frame=2))
    with pytest.raises(vpt.VptError, match="VPT_PIPELINE_WHOLE"):
        g.render(1)
    g.set_params(vpt.default_params(max_depth=4))
    g.render(1)                                          # the same context with one sample per frame: fine
    g.close()
    g = vpt.PathTracer(64, 36)                           # AUTO with two samples per frame: the per-bounce kernels
    g.set_scene(scenes("cornell_box")); g.set_params(vpt.default_params(max_depth=4, samples_per_frame=2))
    g.render(1)
    assert g.stats()["kernel_launches"]["bounce"] > 0
    g.close()


@pytest.mark.parametrize("name,depth", [("cornell_box", 8), ("cornell_box", 200), ("cornell_glass", 40)])
def test_async_frames_are_whole_path_launches(vpt, oracle, scenes, name, depth):
    """vpt_render_async, one frame per call: with the whole-path launch every such batch is a fixed schedule — max_depth = 200 included,
    which the per-bounce pipeline could only enqueue partially — dealt to the lanes and replayed from captured graphs."""
    sc, w, h, frames = scene_of(vpt, scenes, name), 128, 72, 10
    p = vpt.default_params(max_depth=depth)
    ref, _ = oracle_image(oracle, sc, w, h, p, frames)
    g = vpt.PathTracer(w, h, frames_in_flight=1)
    g.set_scene(sc); g.set_params(p)
    prev = 0
    for _ in range(frames):
        done, _t = g.render_async(1)
        assert not done
        cur = g.postprocess_device()
        if prev:
            g.wait(prev)
        prev = cur
    g.wait()
    st = g.stats()
    assert np.array_equal(g.radiance(), ref)
    assert st["kernel_launches"]["primary"] == frames and st["kernel_launches"]["bounce"] == 0 and st["kernel_launches"]["resolve"] == frames
    assert st["graph_launches"] >= frames - 3
    out8, _ = oracle.postprocess(ref, vpt.default_post_params())
    assert np.array_equal(g.output_to_host(), out8)
    g.close()


def test_whole_path_contexts_hold_per_sample_buffers_only(vpt, oracle, scenes):
    """A whole-path batch keeps its paths in registers: the context allocates 48 B per sample (frame sum, medium state) and the records of ONE
    frame instead of ~290 B per resident path — a 32-frame batch at 1080p in < 5 GB (the per-bounce pipelines: 25 GB).  When the parameters
    stop qualifying (two samples per frame) the buffers are replaced before the next batch, and the images stay the oracle's."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")

    def free_bytes():
        f, t = C.c_size_t(0), C.c_size_t(0)
        assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value
    sc = scenes("cornell_box")
    free0 = free_bytes()
    g = vpt.PathTracer(1920, 1080, frames_in_flight=32)
    g.set_scene(sc); g.set_params(vpt.default_params(max_depth=4))
    g.render(32)
    st = g.stats()
    used = free0 - free_bytes()
    assert st["frames_allocated"] == 32 and st["resident_frames"] == 1 and st["kernel_launches"]["bounce"] == 0
    assert used < 5 * 1024 ** 3, "a 32-frame whole-path batch holds %.1f GB" % (used / 2 ** 30)
    g.close()
    # the same context type at a size the oracle does in seconds: whole-path batches, then two samples per frame (per-bounce kernels, records for every resident path)
    w, h = 96, 54
    g = vpt.PathTracer(w, h, frames_in_flight=6)
    g.set_scene(sc)
    p = vpt.default_params(max_depth=5)
    g.set_params(p)
    g.render(6)
    assert g.stats()["resident_frames"] == 1
    o = oracle.Oracle(sc, w, h); o.set_params(p); o.render(6); ref = o.radiance(); o.close()
    assert np.array_equal(g.radiance(), ref)
    p2 = vpt.default_params(max_depth=5, samples_per_frame=2)
    g.set_params(p2); g.reset()
    g.render(6)
    st = g.stats()
    assert st["resident_frames"] == 6 and st["kernel_launches"]["bounce"] > 0
    o = oracle.Oracle(sc, w, h); o.set_params(p2); o.render(6); ref2 = o.radiance(); o.close()
    assert np.array_equal(g.radiance(), ref2)
    g.close()
