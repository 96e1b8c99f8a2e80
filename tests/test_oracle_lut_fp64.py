"""Which side is right where the oracle and the shipped refraction tables disagree?

The refraction tables (Assets/LookupTables/RefractionLookup*.bin) and the oracle's restatement of LookupRefract.slang agree
to Monte-Carlo error everywhere except a corner: near-mirror roughness (rows y <= 4, alpha <= 0.032) seen at grazing angles
(columns x <= 31, V.z <= 0.06), where the shipped values sit up to 0.09 BELOW the oracle's, and layer z = 0 (IOR 1.0001).
This test evaluates the same cells a third way — float64 numpy, vectorised, using the closed form the shader reduces to
(every valid sample contributes BSDF/PDF = G1(L), LookupRefract.slang:53-102 with Material.slang:331-404) — and pins the
outcome: float64 agrees with the fp32 ORACLE in the corner; the shipped table is the outlier there (the reference generated
it through the Vulkan driver's own normalize / sqrt / divide, which are not correctly rounded).  BASELINE config 5 (roughness
0.05, IOR 1.5) reads rows 5-6 of layers 15-16, outside the corner, where all three agree."""
import numpy as np
import pytest


def fp64_cell(x, y, z, above, n=2_000_000, seed=5):
    vc = float(np.clip((x / 127.0) ** 2, 0.01, 0.9999))
    a = float(np.clip(y / 127.0, 0.01, 1.0))
    ior = 1.0 + float(np.clip(z / 31.0, 0.0001, 1.0))
    eta = 1.0 / ior if above else ior
    u = np.random.default_rng(seed).random((n, 4))
    phi_v = 2 * np.pi * u[:, 0]
    mag = np.sqrt(1 - vc * vc)
    V = np.stack([mag * np.cos(phi_v), mag * np.sin(phi_v), np.full(n, vc)], 1)
    # GGX VNDF sample (Sampler.slang:141-166)
    Vh = np.stack([a * V[:, 0], a * V[:, 1], V[:, 2]], 1); Vh /= np.linalg.norm(Vh, axis=1, keepdims=True)
    T1 = np.stack([-Vh[:, 1], Vh[:, 0], np.zeros(n)], 1) / np.sqrt(Vh[:, 0] ** 2 + Vh[:, 1] ** 2)[:, None]
    T2 = np.cross(Vh, T1)
    r, ph = np.sqrt(u[:, 1]), 2 * np.pi * u[:, 2]
    t1, t2 = r * np.cos(ph), r * np.sin(ph)
    s = 0.5 * (1 + Vh[:, 2])
    t2 = (1 - s) * np.sqrt(1 - t1 * t1) + s * t2
    Nh = t1[:, None] * T1 + t2[:, None] * T2 + np.sqrt(np.maximum(0, 1 - t1 * t1 - t2 * t2))[:, None] * Vh
    Hm = np.stack([a * Nh[:, 0], a * Nh[:, 1], np.maximum(0, Nh[:, 2])], 1); Hm /= np.linalg.norm(Hm, axis=1, keepdims=True)
    vh = (V * Hm).sum(1)
    # DielectricFresnel (Material.slang:441-458)
    ci = np.abs(vh)
    st2 = eta * eta * (1 - ci * ci)
    ct = np.sqrt(np.maximum(1 - st2, 0))
    rs, rp = (eta * ct - ci) / (eta * ct + ci), (eta * ci - ct) / (eta * ci + ct)
    F = np.where(st2 > 1, 1.0, 0.5 * (rs * rs + rp * rp))
    refl = u[:, 3] < F

    def g1(L):  # GGXSmithAnisotropic (Material.slang:421-437), isotropic
        return 1 / (1 + (-1 + np.sqrt(1 + a * a * (L[:, 0] ** 2 + L[:, 1] ** 2) / L[:, 2] ** 2)) / 2)
    Lr = 2 * vh[:, None] * Hm - V
    k = 1 - eta * eta * (1 - vh * vh)
    Lt = -eta * V - (-eta * vh + np.sqrt(np.maximum(k, 0)))[:, None] * Hm
    val = np.zeros(n)
    okr = refl & (Lr[:, 2] > 1e-5)
    okt = ~refl & (k >= 0) & (Lt[:, 2] < 0)
    val[okr] = g1(Lr[okr]); val[okt] = g1(Lt[okt])
    return float(val.mean())


CORNER = [(3, 0, 16), (12, 1, 1), (10, 2, 31), (16, 0, 16)]          # near-mirror AND grazing: the disputed cells
CONFIG5 = [(40, 5, 15), (90, 6, 16), (127, 6, 15), (20, 6, 16)]       # rows / layers BASELINE config 5's glass reads
BODY = [(64, 64, 16), (30, 20, 8), (100, 9, 25)]


@pytest.mark.parametrize("kind", [1, 2])
def test_fp64_sides_with_the_oracle_where_the_shipped_table_disagrees(oracle, vpt, kind):
    table = vpt.scenes.load_luts()[kind]
    cells = CORNER + CONFIG5 + BODY
    idx = np.array([x + y * 128 + z * 128 * 128 for x, y, z in cells], np.uint32)
    orc = oracle.lut_cells(kind, (128, 128, 32), 100000, 7, idx)
    for (x, y, z), o in zip(cells, orc):
        ref = fp64_cell(x, y, z, above=(kind == 1))
        ship = float(table[z, y, x])
        assert abs(o - ref) < 0.006, ("oracle vs fp64", kind, x, y, z, o, ref)             # 100 k fp32 samples vs 2 M fp64 samples
        if (x, y, z) in CORNER:
            assert ref - ship > 0.03, ("the shipped table is the outlier here", kind, x, y, z, ship, ref)
        else:
            assert abs(ship - ref) < 0.008, ("shipped vs fp64", kind, x, y, z, ship, ref)
