"""Bloom chain + tonemap kernels against the oracle restatement: RGBA8 output byte-exact, bloom mip 0 bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def hdr_image(w, h, seed):
    rng = np.random.RandomState(seed)
    img = np.zeros((h, w, 4), np.float32)
    img[..., :3] = rng.gamma(0.4, 4.0, (h, w, 3))
    img[rng.rand(h, w) < 0.01, :3] *= 200.0  # fireflies above the bloom threshold
    img[..., 3] = 1.0
    return img


@pytest.mark.parametrize("w,h,kw", [
    (256, 144, {}), (301, 173, {}), (64, 64, dict(mip_count=1)), (97, 33, dict(mip_count=4, bloom_strength=0.6)),
    (128, 72, dict(exposure=2.5, gamma=1.8, bloom_threshold=0.5, falloff_range=0.25)), (2, 2, {}), (5, 3, {}),
])
def test_postprocess_matches_oracle(vpt, oracle, scenes, w, h, kw):
    img = hdr_image(w, h, w + h)
    pp = vpt.default_post_params(**kw)
    g = vpt.PathTracer(w, h)
    g.set_radiance(img, 1)
    out8, bloom = g.postprocess(pp, want_bloom=True)
    g.close()
    ref8, refb = oracle.postprocess(img, pp)
    assert np.array_equal(bloom, refb)
    assert np.array_equal(out8, ref8)


def test_nearest_bloom_tap_variant(vpt, oracle):
    img = hdr_image(120, 68, 3)
    flags = vpt._abi.FLAGS_DEFAULT & ~vpt._abi.FLAG_TONEMAP_LINEAR_BLOOM_TAP
    g = vpt.PathTracer(120, 68)
    g.set_params(vpt.default_params(flags=flags))
    g.set_radiance(img, 1)
    out8 = g.postprocess()
    g.close()
    ref8, _ = oracle.postprocess(img, vpt.default_post_params(), flags)
    assert np.array_equal(out8, ref8)


def test_full_hd_post_properties(vpt, oracle):
    """BASELINE size 1920x1080: exact vs oracle (the scalar restatement finishes in seconds) + invariants."""
    img = hdr_image(1920, 1080, 1)
    g = vpt.PathTracer(1920, 1080)
    g.set_radiance(img, 1)
    out8, bloom = g.postprocess(want_bloom=True)
    st = g.stats()
    g.close()
    ref8, refb = oracle.postprocess(img, vpt.default_post_params())
    assert np.array_equal(out8, ref8) and np.array_equal(bloom, refb)
    assert (out8[..., 3] == 255).all() and np.isfinite(bloom).all() and (bloom[..., :3] >= 0).all()
    assert st["kernel_launches"]["bloom"] == 1 + 9 + 9 and st["kernel_launches"]["tonemap"] == 1  # PostProcessor.cpp:204-245
