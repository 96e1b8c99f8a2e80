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


def cloud_grid(shape=(40, 36, 48), seed=2):
    """A smooth procedural 'smoke' density [z, y, x] (sum of Gaussian puffs), every axis >= 32 voxels so that the
    reference's 32^3 block-maxima table has no holes (it is filled through `x * 32 / dim`, PathTracer.cpp:1439)."""
    rng = np.random.default_rng(seed)
    z, y, x = np.meshgrid(*[np.linspace(-1, 1, n) for n in shape], indexing="ij")
    d = np.zeros(shape, np.float64)
    for _ in range(7):
        c = rng.uniform(-0.6, 0.6, 3); r = rng.uniform(0.2, 0.5)
        d += rng.uniform(0.5, 2.0) * np.exp(-((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) / (r * r))
    return d.astype(np.float32)


def test_heterogeneous_slab_beer_lambert_and_furnace(vpt, oracle):
    """Dense-grid volumes (the reference's NanoVDB path, densified): a constant grid behaves like the homogeneous box,
    a half-filled grid like half the thickness, and an albedo-1 cloud in a white env stays white."""
    sc = empty_scene(vpt)

    def centre(grid, sigma, color, frames):
        o = oracle.Oracle(sc, 64, 36)
        o.set_params(vpt.default_params(max_depth=64, max_samples=1 << 30))
        gi = o.add_density_grid(grid)
        o.set_volumes([vpt.volume(corner_min=(-1, -1, -1), corner_max=(1, 1, 1), color=color, density=sigma, density_data_index=gi)])
        o.render(frames)
        img = o.radiance(); o.close()
        return img[16:20, 30:34, :3].mean()
    tol = 4 * np.sqrt(0.25 / (16 * 256)) + 0.01   # MC error + the epsilon steps between blocks (Volume.slang:124)
    assert abs(centre(np.full((32, 32, 32), 3.0, np.float32), 0.7, (0, 0, 0), 256) - np.exp(-0.7 * 2)) < tol
    half = np.zeros((64, 32, 32), np.float32); half[:32] = 2.0
    assert abs(centre(half, 0.7, (0, 0, 0), 256) - np.exp(-0.7 * 1)) < tol + 0.01   # + the +-1 voxel jitter at the interface
    assert abs(centre(cloud_grid(), 3.0, (1, 1, 1), 96) - 1.0) < 0.03
