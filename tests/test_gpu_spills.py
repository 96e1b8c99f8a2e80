"""The traversal stacks' SPILL path.  A lane's stack is 14 LDS entries; deeper entries go to a per-thread region in global memory, and
the shadow kernels of bounce k run on a second stream beside the extend kernel of bounce k + 1, each grid with a spill region of its
own (vpt_api.hip stack_overflow2; round 2 shared one region between them: a race only a spilling scene can show).  No shipped scene
spills, so this one is built to: 40,000 triangles in 20,000 stacked sheets that every ray from the floor below crosses — a closest-hit
search pushes up to three siblings per level on its way down to the nearest sheet, an any-hit search towards the light above likewise.
vpt_stats.stack_spills counts the spill words the product kernels wrote (per region), so the test knows the path was taken."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def stacked_sheets(vpt, sheets=20000):
    S = vpt.scenes
    sc = S.Scene()
    sc.luts = S.load_luts()
    z = np.linspace(-1.5, 1.5, sheets).astype(np.float32)
    corners = np.array([[-1, -1], [1, -1], [1, 1], [-1, 1]], np.float32)
    pos = np.zeros((sheets, 4, 3), np.float32)
    pos[:, :, :2] = corners[None]
    pos[:, :, 2] = z[:, None]
    idx = (np.arange(sheets, dtype=np.uint32)[:, None] * 4 + np.array([0, 1, 2, 0, 2, 3], np.uint32)[None]).reshape(-1)
    nrm = np.tile(np.array([0, 0, -1], np.float32), (sheets * 4, 1))
    m_stack = sc.add_mesh(pos.reshape(-1, 3), nrm, np.zeros((sheets * 4, 2), np.float32), idx)

    def quad(zq, half, normal_z):
        p = np.array([[-half, -half, zq], [half, -half, zq], [half, half, zq], [-half, half, zq]], np.float32)
        order = [0, 1, 2, 0, 2, 3] if normal_z > 0 else [0, 2, 1, 0, 3, 2]
        return sc.add_mesh(p, np.tile(np.array([0, 0, normal_z], np.float32), (4, 1)), np.zeros((4, 2), np.float32), np.array(order, np.uint32))
    m_floor, m_light = quad(-4.5, 6.0, 1.0), quad(4.0, 3.0, -1.0)
    sc.materials.append(S.material(base_color=(0.7, 0.7, 0.65)))
    sc.materials.append(S.material(base_color=(0.2, 0.5, 0.8), roughness=0.6))
    sc.materials.append(S.material(base_color=(1, 1, 1), emissive_color=(40, 36, 30)))
    sc.add_instance(m_floor, 0); sc.add_instance(m_stack, 1); sc.add_instance(m_light, 2)
    # the camera sits between the floor and the stack, looking down at the floor: every NEE ray goes up through the sheets
    sc.view_inverse = np.linalg.inv(S.look_at((0.0, 0.0, -1.7), (0.0, 0.0, -4.5), (0.0, 1.0, 0.0))).astype(np.float32)
    return sc


def test_spilling_stacks_with_the_two_stream_schedule(vpt, oracle):
    sc = stacked_sheets(vpt)
    w, h, frames = 96, 54, 3
    p = vpt.default_params(max_depth=4)
    o = oracle.Oracle(sc, w, h)
    o.set_params(p); o.render(frames)
    ref = o.radiance(); o.close()
    assert ref[..., :3].max() > 0 and (ref[..., :3].sum(axis=2) == 0).any(), "the fixture should have lit and shadowed floor"

    def run(**kw):
        g = vpt.PathTracer(w, h, **kw)
        g.set_scene(sc); g.set_params(p)
        g.render(frames)
        img, st = g.radiance(), g.stats()
        g.close()
        return img, st

    img_overlap, st_overlap = run()                 # default: shadow kernels + join on the second stream beside the next extend
    img_serial, st_serial = run(profile=True)       # one kernel at a time (timing mode), one stream
    assert st_overlap["kernel_launches"]["join"] > 0, "the streams pipeline should run this scene"
    assert st_overlap["stack_spills"][0] > 0 and st_overlap["stack_spills"][1] > 0, "no spills: %r" % (st_overlap["stack_spills"],)
    assert st_serial["stack_spills"][0] > 0 and st_serial["stack_spills"][1] == 0
    assert np.array_equal(img_overlap, img_serial)
    assert np.array_equal(img_overlap, ref)
    # the other pipelines share the spill path (TravStack in traverse.hpp): fused per-bounce kernel, round 1's stage kernels
    for pipeline in (1, 3) if vpt.has_lab() else (1,):
        img, st = run(pipeline=pipeline)
        assert st["stack_spills"][0] > 0
        assert np.array_equal(img, ref)


def test_shipped_scenes_do_not_spill(vpt, scenes):
    g = vpt.PathTracer(96, 54)
    g.set_scene(scenes("cornell_box_glass")); g.set_params(vpt.default_params(max_depth=8))
    g.render(4)
    assert g.stats()["stack_spills"] == [0, 0]
    g.close()
