"""Lookup-table generator: vpt_lut_calculate vs the oracle's pass-by-pass restatement of
LookupTableCalculator::CalculateTable (bit-exact), and vs the tables the reference ships (Monte-Carlo error)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KINDS = [(0, (64, 64, 32)), (1, (128, 128, 32)), (2, (128, 128, 32))]


@pytest.mark.parametrize("kind,size", KINDS)
def test_lut_bit_exact_vs_oracle(vpt, oracle, kind, size):
    n = size[0] * size[1] * size[2]
    got = vpt.calculate_lut(kind, size, 200, time_ms=5).reshape(-1)   # 10 passes of 20 samples
    rng = np.random.default_rng(10 + kind)
    corners = [x + y * size[0] + z * size[0] * size[1] for x in (0, size[0] - 1) for y in (0, size[1] - 1) for z in (0, size[2] - 1)]
    cells = np.concatenate([np.array(corners), rng.integers(0, n, 6000)]).astype(np.uint32)
    ref = oracle.lut_cells(kind, size, 200, 5, cells)
    assert np.array_equal(got[cells].view(np.uint32), ref.view(np.uint32)), int((got[cells] != ref).sum())
    # a different time seed gives a different table; a sample count that is not a multiple of 20 drops the remainder
    assert not np.array_equal(got, vpt.calculate_lut(kind, size, 200, time_ms=6).reshape(-1))
    assert np.array_equal(vpt.calculate_lut(kind, (16, 16, 4), 219, time_ms=1), vpt.calculate_lut(kind, (16, 16, 4), 219, time_ms=1))


@pytest.mark.parametrize("kind,size", KINDS)
def test_lut_reproduces_shipped_tables(vpt, kind, size):
    """Application.cpp:41,54,67 generated the shipped tables with 10'000'000 samples per cell and clock-derived
    seeds; 200k samples per cell here must agree to Monte-Carlo error."""
    shipped = vpt.scenes.load_luts()[kind]
    got = vpt.calculate_lut(kind, size, 200000, time_ms=3)
    assert got.shape == shipped.shape
    err = np.abs(got - shipped)
    assert np.isfinite(got).all()
    # Rows of near-mirror roughness (refract tables: y < 8, roughness <= 0.055) are limited by fp32 cancellation in
    # the VNDF sample and the GGX D term, not by sample count: this generator is reproducible there to 1e-3 between
    # seeds yet sits up to 0.08 away from the shipped table, which was computed with the Vulkan driver's own
    # sin/cos/sqrt/fma choices.  Everything else agrees to Monte-Carlo error.
    y0 = 8 if kind else 0
    body, mirror = err[:, y0:, :], err[:, :y0, :]
    assert body.mean() < 1.5e-3 and np.quantile(body, 0.999) < 0.02, (body.mean(), float(np.quantile(body, 0.999)), body.max())
    if y0:
        assert mirror.mean() < 0.02 and mirror.max() < 0.15, (mirror.mean(), mirror.max())


def test_lut_argument_errors(vpt):
    with pytest.raises(vpt.VptError):
        vpt.calculate_lut(0, (8, 8, 2), 19)       # fewer than one 20-sample pass
    with pytest.raises(vpt.VptError):
        vpt.calculate_lut(3, (8, 8, 2), 200)      # unknown table kind
    with pytest.raises(vpt.VptError):
        vpt.calculate_lut(0, (0, 8, 2), 200)
