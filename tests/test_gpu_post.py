"""Bloom chain + tonemap kernels against the oracle restatement: RGBA8 output byte-exact, bloom mip 0 bit-exact — for BOTH
schedules of vpt_postprocess: the fused one (default: threshold inside the first down-sample, small mips down and up in one
launch, last up-sample + tonemap in one kernel) and the reference's passes one kernel each (vpt_post_params.schedule = 1)."""
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
@pytest.mark.parametrize("schedule", [0, 1])
def test_postprocess_matches_oracle(vpt, oracle, scenes, w, h, kw, schedule):
    img = hdr_image(w, h, w + h)
    pp = vpt.default_post_params(schedule=schedule, **kw)
    g = vpt.PathTracer(w, h)
    g.set_radiance(img, 1)
    out8, bloom = g.postprocess(pp, want_bloom=True)
    only8 = g.postprocess(pp)          # the fused schedule skips the mip-0 store when nobody asks for it
    g.close()
    ref8, refb = oracle.postprocess(img, pp)
    assert np.array_equal(bloom, refb)
    assert np.array_equal(out8, ref8) and np.array_equal(only8, ref8)


@pytest.mark.parametrize("w,h", [(1920, 1080), (3840, 2160), (1001, 563), (130, 70), (66, 5), (4, 64)])
def test_fused_schedule_equals_reference_passes(vpt, w, h):
    """The two schedules against each other at sizes the scalar oracle would take long for (incl. 4K, odd sizes, tiles that
    straddle the image edge, images narrower than a tile): RGBA8 and bloom mip 0 identical, with both bloom taps."""
    img = hdr_image(w, h, 7)
    for flags in (vpt._abi.FLAGS_DEFAULT, vpt._abi.FLAGS_DEFAULT & ~vpt._abi.FLAG_TONEMAP_LINEAR_BLOOM_TAP):
        g = vpt.PathTracer(w, h)
        g.set_params(vpt.default_params(flags=flags))
        g.set_radiance(img, 1)
        a8, ab = g.postprocess(vpt.default_post_params(schedule=0), want_bloom=True)
        b8, bb = g.postprocess(vpt.default_post_params(schedule=1), want_bloom=True)
        g.close()
        assert np.array_equal(a8, b8) and np.array_equal(ab, bb)


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
    # fused: first down-sample (thresholds), down-samples 2 -> 3 -> 4 in one launch, tail (mips 5-9 down and up), the three up-samples
    # 4 -> 3 -> 2 -> 1 in one launch, then last up-sample + tonemap in one
    assert st["kernel_launches"]["bloom"] == 1 + 1 + 1 + 1 and st["kernel_launches"]["tonemap"] == 1
    g = vpt.PathTracer(1920, 1080)
    g.set_radiance(img, 1)
    p8 = g.postprocess(vpt.default_post_params(schedule=1))
    st = g.stats()
    g.close()
    assert np.array_equal(p8, ref8)
    assert st["kernel_launches"]["bloom"] == 1 + 9 + 9 and st["kernel_launches"]["tonemap"] == 1  # PostProcessor.cpp:204-245
