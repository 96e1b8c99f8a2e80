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
    # The refraction tables disagree with this generator in one corner only: near-mirror rows (y <= 4, alpha <= 0.032) seen
    # at grazing angles (x <= 31, V.z <= 0.06) — up to 0.09 — and layer z = 0 (IOR 1.0001).  A float64 evaluation of those cells sides
    # with this generator (tests/test_oracle_lut_fp64.py): the shipped values are the outlier, so the corner is excluded
    # and everything else, the other near-mirror cells included, has to agree to Monte-Carlo error.
    disputed = np.zeros(err.shape, bool)
    if kind:
        disputed[:, :5, :32] = True
        disputed[0, :, :] = True
    ok = err[~disputed]
    assert ok.mean() < 1.5e-3 and np.quantile(ok, 0.999) < 0.012, (ok.mean(), float(np.quantile(ok, 0.999)), ok.max())
    if kind:
        mirror = np.where(disputed, 0.0, err)[1:, :8, :]                     # near-mirror rows outside the corner: the rows config 5 reads
        assert mirror.max() < 0.012, mirror.max()
        assert (got - shipped)[1:, :2, :13].mean() > 0.05   # in the corner the shipped table is LOW by ~0.08


def test_lut_argument_errors(vpt):
    with pytest.raises(vpt.VptError):
        vpt.calculate_lut(0, (8, 8, 2), 19)       # fewer than one 20-sample pass
    with pytest.raises(vpt.VptError):
        vpt.calculate_lut(3, (8, 8, 2), 200)      # unknown table kind
    with pytest.raises(vpt.VptError):
        vpt.calculate_lut(0, (0, 8, 2), 200)
