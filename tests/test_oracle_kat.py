"""Pins the oracle against everything the reference tree offers for this path:
   * Assets/LookupTables/*.bin — produced by the reference's own Material/Sampler code; a Monte-Carlo
     run of LookupReflect/LookupRefract.slang through the ORACLE's BSDF functions must reproduce cells;
   * furnace mode (Material.slang:78-86, Miss.slang:61-65): image == 1 +- MC error;
   * analytic checks on the env alias table and on the oracle's own BVH."""
import numpy as np
import pytest


def test_reflection_lut_cells(oracle, vpt):
    lut = vpt.scenes.load_luts()[0]  # [z,y,x] 32x64x64
    L = oracle.lib()
    worst = 0.0
    for (x, y, z) in [(32, 32, 0), (10, 50, 0), (60, 8, 16), (20, 20, 31), (63, 63, 8), (5, 60, 4)]:
        mc = L.orc_lut_reflect_cell(x, y, z, 64, 64, 32, 400000, 1234)
        worst = max(worst, abs(mc - float(lut[z, y, x])))
        assert abs(mc - float(lut[z, y, x])) < 4e-3, (x, y, z, mc, float(lut[z, y, x]))
    # Reflect[z=0,y=32,x=32] = 0.6855
    assert abs(float(lut[0, 32, 32]) - 0.6855) < 2e-3


@pytest.mark.parametrize("above", [1, 0])
def test_refraction_lut_cells(oracle, vpt, above):
    luts = vpt.scenes.load_luts()
    lut = luts[1] if above else luts[2]
    L = oracle.lib()
    for (x, y, z) in [(64, 64, 16), (127, 127, 31), (30, 100, 8), (100, 20, 24), (16, 40, 2)]:
        mc = L.orc_lut_refract_cell(x, y, z, 128, 128, 32, above, 400000, 99)
        assert abs(mc - float(lut[z, y, x])) < 5e-3, (above, x, y, z, mc, float(lut[z, y, x]))


@pytest.mark.parametrize("label,kw,lo,hi", [
    # energy-compensated glass: the refraction tables are exactly E[f cos / pdf], so the furnace is white
    ("glass", dict(transmission=1.0, roughness=0.3), 0.99, 1.01),
    # energy-compensated metal: white up to the rejected (below-horizon) samples, ~1 % per bounce
    ("metal", dict(metallic=1.0, roughness=0.2), 0.97, 1.01),
    # as a BSDF the dielectric blend integrates to 1 (tests/test_oracle_bsdf_fp64.py, float64); the furnace still loses energy
    # (0.83 inside the box) because the reference's estimator divides by a pdf that is not its sampling density (lobe chosen
    # with F(V.H_sampled), evaluated with F(V.H_(V+L)); below-horizon draws rejected) — bounded so a regression shows up
    ("dielectric", dict(), 0.78, 0.90),
])
def test_furnace_mode(oracle, vpt, scenes, label, kw, lo, hi):
    """FURNACE_TEST_MODE (Material.slang:78-86, Miss.slang:61-65): albedos 1, emission 0, env 1."""
    import copy
    sc = copy.deepcopy(scenes("cornell_box"))
    # SampleEmissiveTriangle reads the raw material emission (Sampler.slang:417), which furnace mode does
    # not override, so the quad light must be switched off for the white-furnace identity to hold.
    for m in sc.materials:
        m["emissive_color"] = (0.0, 0.0, 0.0)
        m.update(kw)
    a = vpt._abi
    o = oracle.Oracle(sc, 48, 27)
    o.set_params(vpt.default_params(max_depth=64, flags=a.FLAGS_DEFAULT | a.FLAG_FURNACE, max_luminance=1e9))
    o.render(64)
    img = o.radiance()[..., :3]
    assert np.isfinite(img).all()
    inside = float(img[5:22, 14:34].mean())  # pixels that look into the box
    assert lo < inside < hi, (label, inside)
    assert float(img[:, :6].mean()) == 1.0  # primary rays that miss see the furnace directly
    o.close()


def test_energy_compensation_matters(oracle, vpt, scenes):
    import copy
    sc = copy.deepcopy(scenes("cornell_box"))
    for m in sc.materials:
        m["emissive_color"] = (0.0, 0.0, 0.0)
        m.update(dict(metallic=1.0, roughness=0.6))
    a = vpt._abi
    o = oracle.Oracle(sc, 48, 27)
    o.set_params(vpt.default_params(max_depth=64, flags=(a.FLAGS_DEFAULT | a.FLAG_FURNACE) & ~a.FLAG_ENERGY_COMPENSATION, max_luminance=1e9))
    o.render(16)
    assert float(o.radiance()[5:22, 14:34, :3].mean()) < 0.4  # without the LUTs rough metal loses most energy
    o.close()


def test_env_alias_table_reproduces_texel_distribution(oracle, vpt, scenes):
    sc = scenes("cornell_box")
    import copy
    sc = copy.copy(sc)
    sc.env = vpt.scenes.sun_sky_env(32, 16, seed=5, sun_peak=200.0)
    o = oracle.Oracle(sc, 8, 8)
    n = 32 * 16
    alias = np.zeros(n, np.uint32); imp = np.zeros(n, np.float32); pdf = np.zeros(n, np.float32)
    oracle.lib().orc_get_env_tables(o.h_, alias.ctypes.data, imp.ctypes.data, pdf.ctypes.data)
    # probability of landing on texel j through the alias table
    p = np.zeros(n)
    for i in range(n):
        q = min(max(float(imp[i]), 0.0), 1.0)
        p[i] += q / n
        p[alias[i]] += (1.0 - q) / n
    e = sc.env[..., :3].max(axis=2).reshape(-1).astype(np.float64)
    theta = (np.arange(17) * np.pi / 16)
    area = np.repeat((np.cos(theta[:-1]) - np.cos(theta[1:])) * (2 * np.pi / 32), 32)
    target = e * area / (e * area).sum()
    assert abs(p.sum() - 1.0) < 1e-5
    # Upstream's partition pre-increments its cursor (PathTracer.cpp:1244-1248): slot 0 of
    # the low list is never written and the last low texel is dropped, so the table is only approximately
    # the target distribution.  The quirk is reproduced, hence a total-variation bound rather than equality.
    tv = 0.5 * np.abs(p - target).sum()
    assert tv < 0.01, tv
    # alpha = pdf per steradian: sum(pdf * solid angle) == 1
    assert abs((pdf.astype(np.float64) * area).sum() - 1.0) < 1e-4
    o.close()


def test_black_env_tables(oracle, scenes):
    """sum == 0: importance 0, alias = self, pdf 0 (PathTracer.cpp:1214-1218, 1287-1290)."""
    o = oracle.Oracle(scenes("cornell_box"), 8, 8)
    alias = np.zeros(1, np.uint32); imp = np.ones(1, np.float32); pdf = np.ones(1, np.float32)
    oracle.lib().orc_get_env_tables(o.h_, alias.ctypes.data, imp.ctypes.data, pdf.ctypes.data)
    assert alias[0] == 0 and imp[0] == 0 and pdf[0] == 0
    o.close()


def random_rays(n, seed, scale=12.0):
    rng = np.random.RandomState(seed)
    r = np.zeros((n, 8), np.float32)
    r[:, 0:3] = (rng.rand(n, 3) * 2 - 1) * scale
    d = rng.randn(n, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r[:, 4:7] = d
    r[:, 3] = 1e-4
    r[:, 7] = 1e6
    return r


@pytest.mark.parametrize("name", ["cornell_box", "cornell_box_glass", "viking_room"])
def test_oracle_bvh_equals_brute_force(oracle, scenes, name):
    sc = scenes(name)
    o = oracle.Oracle(sc, 8, 8)
    rays = random_rays(20000 if name != "viking_room" else 4000, 11, scale=6.0 if name != "viking_room" else 2.0)
    a = o.trace_rays(rays)
    o.set_brute_force(True)
    b = o.trace_rays(rays)
    assert (a["t"] >= 0).sum() > len(rays) // 20
    for k in ("t", "u", "v", "primitive", "instance"):
        assert np.array_equal(a[k], b[k]), k
    o.close()


def test_emissive_list_and_scene_info(oracle, scenes):
    o = oracle.Oracle(scenes("cornell_box"), 8, 8)
    info = o.scene_info()
    assert info["tris"] == 12 and info["emissive_meshes"] == 1 and info["emissive_tris"] == 2  # SURVEY §4 fixtures
    o.close()


def test_oracle_render_is_deterministic_and_seeded(oracle, vpt, scenes):
    sc = scenes("cornell_box")
    imgs = []
    for seed in (1, 1, 2):
        o = oracle.Oracle(sc, 32, 18, threads=3 if seed == 1 else 1)
        o.set_params(vpt.default_params(max_depth=4, base_seed=seed))
        o.render(2)
        imgs.append(o.radiance())
        o.close()
    assert np.array_equal(imgs[0], imgs[1])
    assert not np.array_equal(imgs[0], imgs[2])
    assert (imgs[0][..., 3] == 1.0).all()


@pytest.mark.parametrize("kind,size", [(0, (64, 64, 32)), (1, (128, 128, 32)), (2, (128, 128, 32))])
def test_lut_generator_restatement_reproduces_shipped_tables(oracle, vpt, kind, size):
    """orc_lut_cells restates LookupTableCalculator::CalculateTable pass by pass (20-sample passes, per-pass
    reseed, fp32 sums); at 40k samples a cell lands within Monte-Carlo error of the shipped 10M-sample table."""
    table = vpt.scenes.load_luts()[kind].reshape(-1)
    cells = np.random.default_rng(kind).integers(0, table.size, 96).astype(np.uint32)
    if kind:  # the grazing near-mirror corner and the IOR-1.0001 layer, where a float64 evaluation shows the SHIPPED values
              # to be the outlier (tests/test_oracle_lut_fp64.py), are compared there, not here
        x, y, z = cells % size[0], (cells // size[0]) % size[1], cells // (size[0] * size[1])
        cells = cells[~(((y < 5) & (x < 32)) | (z == 0))]
    got = oracle.lut_cells(kind, size, 40000, 7, cells)
    err = np.abs(got - table[cells])
    assert err.mean() < 4e-3 and err.max() < 0.02, (err.mean(), err.max())
    # the pass structure matters: the same cells with another time seed differ in the low bits but agree statistically
    other = oracle.lut_cells(kind, size, 40000, 8, cells)
    assert not np.array_equal(got, other) and np.abs(got - other).mean() < 6e-3
