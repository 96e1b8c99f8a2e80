"""Pins the oracle's volume restatement (RayGen.slang:162-380, Volume.slang homogeneous branch) with closed forms —
the reference tree holds no golden data for volumes ("parity unpinned" against the reference itself):
Beer-Lambert through a black-albedo slab, and the white furnace (albedo-1 medium in a white env stays white) for
the two phase functions whose sampling density equals their evaluated pdf."""
import numpy as np
import pytest


def empty_scene(vpt, env_rgb=(1, 1, 1)):
    S = vpt.scenes
    s = S.Scene()
    pos = np.array([[0, 0, 500], [0.001, 0, 500], [0, 0.001, 500]], np.float32)  # a speck far behind the camera
    m = s.add_mesh(pos, np.tile([0, 0, 1], (3, 1)).astype(np.float32), np.zeros((3, 2), np.float32), np.array([0, 1, 2], np.uint32))
    s.materials.append(S.material())
    s.add_instance(m, 0)
    s.env = S.constant_env(env_rgb)
    s.luts = S.load_luts()
    s.view_inverse = np.linalg.inv(S.look_at((0, 0, 10), (0, 0, 0), (0, 1, 0))).astype(np.float32)
    return s


def run(vpt, oracle, sc, vols, phase=0, frames=64, w=64, h=36, depth=64):
    o = oracle.Oracle(sc, w, h)
    o.set_params(vpt.default_params(max_depth=depth, max_samples=1 << 30))
    o.set_volumes(vols); o.set_phase_function(phase)
    o.render(frames)
    img = o.radiance(); o.close()
    return img


@pytest.mark.parametrize("sigma,thickness", [(0.7, 2.0), (0.15, 6.0)])
def test_beer_lambert_through_an_absorbing_slab(vpt, oracle, sigma, thickness):
    """Albedo 0: every scatter event kills the path, so a pixel looking straight through sees exp(-sigma d) of the
    (directly visible) white env."""
    sc = empty_scene(vpt)
    slab = vpt.volume(corner_min=(-80, -80, -thickness / 2), corner_max=(80, 80, thickness / 2), color=(0, 0, 0), density=sigma)
    img = run(vpt, oracle, sc, [slab], frames=256)
    c = img[16:20, 30:34, :3].mean()   # 16 central pixels x 256 spp, rays within 1.5 degrees of the slab normal
    expect = np.exp(-sigma * thickness)
    assert abs(c - expect) < 4 * np.sqrt(expect * (1 - expect) / (16 * 256)) + 2e-3, (c, expect)
    # two stacked half-thickness slabs multiply (CalculateVolumesTransmittance is a product over boxes; free-flight
    # sampling draws one distance per box)
    halves = [vpt.volume(corner_min=(-80, -80, -thickness / 2), corner_max=(80, 80, 0), color=(0, 0, 0), density=sigma),
              vpt.volume(corner_min=(-80, -80, 0), corner_max=(80, 80, thickness / 2), color=(0, 0, 0), density=sigma)]
    c2 = run(vpt, oracle, sc, halves, frames=256)[16:20, 30:34, :3].mean()
    assert abs(c2 - expect) < 4 * np.sqrt(expect * (1 - expect) / (16 * 256)) + 2e-3, (c2, expect)


@pytest.mark.parametrize("phase,g", [(0, 0.0), (0, 0.6), (0, -0.5), (1, 0.5)])
def test_white_furnace_in_a_scattering_box(vpt, oracle, phase, g):
    """Albedo 1 in a white env: radiance is 1 along every path, whatever the phase function, if the phase pdf is
    normalised, its sampler matches it, NEE/MIS weights sum to one and transmittance is consistent with sampling."""
    sc = empty_scene(vpt)
    box = vpt.volume(corner_min=(-3, -3, -3), corner_max=(3, 3, 3), color=(1, 1, 1), density=0.8, anisotropy=g, alpha=0.7)
    img = run(vpt, oracle, sc, [box], phase=phase, frames=96)
    inside = img[10:26, 20:44, :3]
    assert abs(inside.mean() - 1.0) < 0.02, inside.mean()
    assert abs(img[..., :3].mean() - 1.0) < 0.01


def test_no_volumes_is_the_plain_integrator(vpt, oracle, scenes):
    """An empty volume list and a volume the camera never looks through leave the image bits unchanged."""
    sc = scenes("cornell_box")
    P = vpt.default_params(max_depth=5)
    o = oracle.Oracle(sc, 64, 36); o.set_params(P); o.render(2); base = o.radiance()
    o.set_volumes([]); o.render(2); assert np.array_equal(base, o.radiance())
    o.set_volumes([vpt.volume(corner_min=(900, 900, 900), corner_max=(901, 901, 901))]); o.render(2)
    far = o.radiance(); o.close()
    assert np.array_equal(base, far)
