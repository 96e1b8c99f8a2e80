"""Oracle post chain vs an independent vectorised numpy restatement of PostProcess/*.slang."""
import numpy as np
import pytest


def mip_sizes(w, h):
    out = []
    for _ in range(10):  # PostProcessor.cpp:136-157
        out.append((w, h))
        w -= w % 2
        h -= h % 2
        w //= 2
        h //= 2
        if w < 2 or h < 2:
            break
    return out


def test_mip_chain_matches_survey():
    assert mip_sizes(1920, 1080) == [(1920, 1080), (960, 540), (480, 270), (240, 135), (120, 67), (60, 33), (30, 16), (15, 8), (7, 4), (3, 2)]


def np_post(img, pp, linear=True):
    f = np.float32
    img = img.astype(f)
    h, w = img.shape[:2]
    sizes = mip_sizes(w, h)
    mc = max(1, min(int(pp.mip_count), len(sizes)))
    rgb = img[..., :3]
    br = (rgb[..., 0] * f(0.2126) + rgb[..., 1] * f(0.7152)) + rgb[..., 2] * f(0.0722)
    e0, e1 = f(pp.bloom_threshold) - f(pp.falloff_range), f(pp.bloom_threshold) + f(pp.falloff_range)
    t = np.clip((br - e0) / (e1 - e0), f(0), f(1))
    mips = [rgb * (t * t * (f(3) - f(2) * t))[..., None]]
    for i in range(1, mc):
        ow, oh = sizes[i]
        src = mips[i - 1]
        ih, iw = src.shape[:2]
        acc = np.zeros((oh, ow, 3), f)
        ys, xs = np.mgrid[0:oh, 0:ow]
        for a in range(-2, 2):
            for b in range(-2, 2):
                acc = acc + src[np.clip(2 * ys + b, 0, ih - 1), np.clip(2 * xs + a, 0, iw - 1)]
        mips.append(acc * (f(1) / f(25)) * f(pp.bloom_strength))  # float3 / float == multiply by the reciprocal (fp32 contract)
    for i in range(mc - 1, 0, -1):
        src, dst = mips[i], mips[i - 1]
        ih, iw = src.shape[:2]
        oh, ow = dst.shape[:2]
        acc = np.zeros((oh, ow, 3), f)
        ys, xs = np.mgrid[0:oh, 0:ow]
        for a in range(-2, 2):
            for b in range(-2, 2):
                acc = acc + src[np.clip(ys // 2 + b + 1, 0, ih - 1), np.clip(xs // 2 + a + 1, 0, iw - 1)]
        mips[i - 1] = acc * (f(1) / f(25)) * f(pp.bloom_strength) + dst
    return mips[0]


@pytest.mark.parametrize("w,h,mips", [(64, 36, 10), (101, 57, 10), (40, 40, 1), (33, 17, 3)])
def test_bloom_chain_matches_numpy(oracle, vpt, w, h, mips):
    rng = np.random.RandomState(w * 7 + h)
    img = np.zeros((h, w, 4), np.float32)
    img[..., :3] = rng.gamma(0.5, 3.0, (h, w, 3))
    img[..., 3] = 1
    pp = vpt.default_post_params(mip_count=mips, bloom_strength=0.8)
    out8, bloom = oracle.postprocess(img, pp)
    ref = np_post(img, pp)
    assert np.array_equal(bloom[..., :3], ref)  # same fp32 operation order -> bit-identical
    assert (bloom[..., 3] == 1).all() and (out8[..., 3] == 255).all()


def test_tonemap_matches_float64_reference_within_one_lsb(oracle, vpt):
    rng = np.random.RandomState(5)
    h, w = 32, 48
    img = np.zeros((h, w, 4), np.float32)
    img[..., :3] = rng.gamma(0.7, 1.0, (h, w, 3))
    pp = vpt.default_post_params(bloom_threshold=1e9, falloff_range=1.0)  # bloom off: threshold far above
    out8, bloom = oracle.postprocess(img, pp)
    assert bloom[..., :3].max() == 0
    c = img[..., :3].astype(np.float64) ** (1 / 2.2)
    m_in = np.array([[0.59719, 0.35458, 0.04823], [0.07600, 0.90834, 0.01566], [0.02840, 0.13383, 0.83777]])
    m_out = np.array([[1.60475, -0.53108, -0.07367], [-0.10208, 1.10813, -0.00605], [-0.00327, -0.07276, 1.07602]])
    a = c @ m_in.T
    r = (a * (a + 0.0245786) - 0.000090537) / (a * (0.983729 * a + 0.4329510) + 0.238081)
    ref = np.clip(r @ m_out.T, 0, 1) * 255
    assert np.abs(out8[..., :3].astype(np.float64) - ref).max() <= 0.5 + 2e-3


def test_threshold_is_smoothstep(oracle, vpt):
    img = np.zeros((4, 8, 4), np.float32)
    lum = np.linspace(-4, 9, 32).reshape(4, 8)
    img[..., :3] = lum[..., None]
    _, bloom = oracle.postprocess(img, vpt.default_post_params(mip_count=1))
    t = np.clip((lum * (0.2126 + 0.7152 + 0.0722) - (-3)) / 10, 0, 1)
    assert np.allclose(bloom[..., 0], lum * (t * t * (3 - 2 * t)), atol=1e-5)
