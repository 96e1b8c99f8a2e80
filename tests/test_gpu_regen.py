"""Path regeneration (vpt_config.resident_frames; kernels_stream.hip k_refill_plan): a batch of F frames with only K < F frames of paths in
flight — behind every shade stage the room the ended paths left in the next ray queue is refilled with the batch's next unstarted samples
(a contiguous block of fresh camera rays).  Seeds depend on (pixel, frame) only and the running mean is applied in frame order after the
batch, so the image must equal the all-resident schedule's and the oracle's bit for bit, for every K, on the pipelines that regenerate
(streams, class-sorted streams), with several samples per frame, across batches, and on row shards; whole-path launches hold no records at
all and the fused per-bounce kernels keep every sample resident: resident_frames changes nothing there."""
import copy

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def oracle_image(oracle, sc, w, h, params, frames):
    o = oracle.Oracle(sc, w, h)
    o.set_params(params)
    o.render(frames)
    ref = o.radiance()
    o.close()
    return ref


@pytest.mark.parametrize("name,pipeline,depth", [("cornell_box", 0, 8), ("cornell_box", 1, 8), ("cornell_box_glass", 0, 12), ("cornell_box_glass", 4, 12), ("cornell_box_glass", 1, 12),
                                                 ("viking_room", 0, 6)])
def test_regenerated_batches_equal_the_oracle(vpt, oracle, scenes, name, pipeline, depth):
    sc, w, h, frames = scenes(name), 96, 54, 13
    p = vpt.default_params(max_depth=depth)
    ref = oracle_image(oracle, sc, w, h, p, frames)
    for K in (1, 3, 5, 13):
        g = vpt.PathTracer(w, h, pipeline=pipeline, frames_in_flight=frames, resident_frames=K)
        g.set_scene(sc); g.set_params(p)
        g.render(frames)
        st = g.stats()
        # what the context holds: K frames of records on the streams; ONE on a whole-path context (cornell_box under AUTO: paths live in registers); all of them on the fused per-bounce kernels
        expect = K if name != "cornell_box" else (1 if pipeline == 0 else frames)
        if pipeline == 1: expect = frames
        assert st["resident_frames"] == expect and st["frames_allocated"] == frames and st["samples"] == w * h * frames
        if name != "cornell_box" and pipeline != 1 and K < frames: assert st["kernel_launches"]["primary"] > 1   # the camera-ray launch + refills
        assert np.array_equal(g.radiance(), ref), (name, pipeline, K)
        g.close()


def test_regeneration_with_samples_per_frame_and_several_batches(vpt, oracle, scenes):
    """samples_per_frame = 3 regenerates inside a slot first (RayGen.slang:33: the RNG stream continues), then across frames; the second
    batch starts where the first ended (dispatch base) and a short last batch fits the resident window entirely."""
    sc, w, h = scenes("cornell_box_glass"), 80, 45
    p = vpt.default_params(max_depth=6, samples_per_frame=3)
    ref = oracle_image(oracle, sc, w, h, p, 11)
    for pipeline in (0, 1):
        g = vpt.PathTracer(w, h, pipeline=pipeline, frames_in_flight=5, resident_frames=2)
        g.set_scene(sc); g.set_params(p)
        g.render(11)          # batches of 5, 5 and 1 frames
        assert np.array_equal(g.radiance(), ref), pipeline
        g.close()


def test_regeneration_on_row_shards_and_after_a_material_edit(vpt, oracle, scenes):
    sc, w, h, frames = copy.deepcopy(scenes("cornell_box_glass")), 64, 37, 9      # 37 rows over 3 shards: ragged; the glass sphere's BVH lives in memory: streams
    p = vpt.default_params(max_depth=5)
    whole = vpt.PathTracer(w, h, frames_in_flight=frames, resident_frames=2)
    whole.set_scene(sc); whole.set_params(p); whole.render(frames)
    img = whole.radiance()
    assert np.array_equal(img, oracle_image(oracle, sc, w, h, p, frames))
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    parts = []
    for r in range(3):
        g = vpt.PathTracer(w, h, shard_rank=r, shard_count=3, frames_in_flight=frames, resident_frames=2)
        g.set_scene(sc); g.set_params(p); g.render(frames)
        assert g.stats()["resident_frames"] == 2
        parts.append(g)
    n = parts[0].shard_floats()
    buf = C.c_void_p()
    assert hip.hipMalloc(C.byref(buf), n * 4 * 3) == 0
    for r, g in enumerate(parts):
        g.shard_to_device(C.c_void_p(buf.value + r * n * 4))
    parts[0].assemble_shards(buf, 3)
    assert np.array_equal(parts[0].radiance(), img)
    hip.hipFree(buf)
    for g in parts:
        g.close()
    m = whole.get_material(2); m.base_color[:] = (0.9, 0.2, 0.1)
    whole.set_material(2, m); whole.render(frames)
    sc.materials[2].update(base_color=(0.9, 0.2, 0.1))
    assert np.array_equal(whole.radiance(), oracle_image(oracle, sc, w, h, p, frames))
    whole.close()


def test_configurations_that_keep_every_sample_resident(vpt, oracle, scenes):
    """Media batches, split-screen dispatch, the fused per-bounce kernels and round 1's stage kernels do not regenerate: resident_frames is ignored there (and the
    buffers grow accordingly), images as before."""
    sc, w, h, frames = scenes("cornell_box_glass"), 64, 36, 6
    p = vpt.default_params(max_depth=5)
    ref = oracle_image(oracle, sc, w, h, p, frames)
    for pipeline in (1, 3) if vpt.has_lab() else (1,):   # the fused per-bounce kernels; VPT_PIPELINE_STAGED_R1 (laboratory build)
        g = vpt.PathTracer(w, h, pipeline=pipeline, frames_in_flight=frames, resident_frames=2)
        g.set_scene(sc); g.set_params(p); g.render(frames)
        assert g.stats()["resident_frames"] == frames and np.array_equal(g.radiance(), ref)
        g.close()
    ps = vpt.default_params(max_depth=5, screen_chunk_count=2)
    o = oracle.Oracle(sc, w, h); o.set_params(ps); o.render(8); ref2 = o.radiance(); o.close()
    g = vpt.PathTracer(w, h, frames_in_flight=8, resident_frames=2)
    g.set_scene(sc); g.set_params(ps); g.render(8)
    assert g.stats()["resident_frames"] == 8 and np.array_equal(g.radiance(), ref2)
    # the same context back on whole-frame dispatches (the larger buffers stay, so this batch is all-resident again)
    g.set_params(p); g.render(frames)
    assert np.array_equal(g.radiance(), ref)
    g.close()
    vol = vpt.volume(corner_min=(-0.6, -1.2, -0.6), corner_max=(0.6, 0.0, 0.6), color=(0.8, 0.8, 0.9), density=1.5)
    o = oracle.Oracle(sc, w, h); o.set_params(p); o.set_volumes([vol]); o.render(frames); ref3 = o.radiance(); o.close()
    g = vpt.PathTracer(w, h, frames_in_flight=frames, resident_frames=2)
    g.set_scene(sc); g.set_params(p); g.set_volumes([vol]); g.render(frames)
    assert g.stats()["resident_frames"] == frames and np.array_equal(g.radiance(), ref3)
    g.close()
