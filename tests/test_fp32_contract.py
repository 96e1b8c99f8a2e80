"""include/vpt_fp32.h: PCG known answers and accuracy of the pinned elementary functions."""
import numpy as np
import pytest


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    ulp = np.maximum(ulp, np.finfo(np.float32).tiny)
    return np.abs(got.astype(np.float64) - ref64) / ulp


def test_pcg_known_answers(oracle):
    L = oracle.lib()
    # derived from Sampler.slang:4-9 == PathTracer.cpp:130-134
    assert [L.orc_pcg_hash(x) for x in (0, 1, 2, 12345, 0xFFFFFFFF)] == [129708002, 2831084092, 2055130248, 4099845390, 3861530882]
    chain, s = [], 0
    for _ in range(4):
        s = L.orc_pcg_hash(s)
        chain.append(s)
    assert chain == [129708002, 817759070, 2145236065, 2368882721]
    fl = [L.orc_uniform_float(c) for c in chain]
    assert np.allclose(fl, [0.0302, 0.19040, 0.49948, 0.55155], atol=5e-5)


def test_uniform_float_range_is_closed_unit_interval(oracle):
    L = oracle.lib()
    assert L.orc_uniform_float(0) == 0.0
    assert L.orc_uniform_float(0xFFFFFFFF) == 1.0  # float(UINT_MAX) rounds to 2^32: inclusive upper end
    assert L.orc_uniform_float(0x80000000) == 0.5


@pytest.mark.parametrize("fn,lo,hi,ref,tol", [
    ("sin", -7.0, 7.0, np.sin, 4.0), ("cos", -7.0, 7.0, np.cos, 4.0), ("log", 1e-6, 1e6, np.log, 4.0),
    ("exp", -20.0, 20.0, np.exp, 4.0), ("asin", -1.0, 1.0, np.arcsin, 4.0), ("acos", -1.0, 1.0, np.arccos, 4.0),
])
def test_elementary_accuracy(oracle, fn, lo, hi, ref, tol):
    rng = np.random.RandomState(3)
    x = (rng.rand(200000) * (hi - lo) + lo).astype(np.float32)
    got = oracle.fp32_eval(fn, x)
    r = ref(x.astype(np.float64))
    if fn in ("sin", "cos"):  # absolute accuracy near zeros of sin/cos, relative elsewhere
        err = np.abs(got - r) / np.maximum(np.spacing(np.abs(r).astype(np.float32)), 6e-8)
    else:
        err = ulp_err(got, r)
    assert err.max() <= tol, (fn, err.max())


def test_atan2_and_pow(oracle):
    rng = np.random.RandomState(4)
    y = (rng.rand(100000) * 4 - 2).astype(np.float32)
    x = (rng.rand(100000) * 4 - 2).astype(np.float32)
    got = oracle.fp32_eval("atan2", y, x)
    assert np.abs(got - np.arctan2(y.astype(np.float64), x.astype(np.float64))).max() < 5e-7
    b = (rng.rand(100000) * 4 + 1e-3).astype(np.float32)
    for e in (2.0, 0.5, 2.2, 1 / 2.2):
        got = oracle.fp32_eval("pow", b, np.full_like(b, e))
        ref = b.astype(np.float64) ** e
        assert (np.abs(got - ref) / ref).max() < 2e-6, e
    # edge semantics used by the path: pow(0,y>0)=0, pow(x,0)=1, pow(<0, non-folded)=NaN, overflow -> inf
    edge = oracle.fp32_eval("pow", np.array([0, 3, -1, 1e20, 1], np.float32), np.array([2.2, 0, 3, 2, 7], np.float32))
    assert edge[0] == 0 and edge[1] == 1 and np.isnan(edge[2]) and np.isinf(edge[3]) and edge[4] == 1
    # constant exponents 2 and 0.5 fold to x*x and sqrt(x) exactly
    x = np.array([3.0, -1.5, 0.3, 1e-20], np.float32)
    assert np.array_equal(oracle.fp32_eval("pow", x, np.full_like(x, 2.0)), x * x)
    assert np.array_equal(oracle.fp32_eval("pow", np.abs(x), np.full_like(x, 0.5)), np.sqrt(np.abs(x)))


def test_special_values(oracle):
    assert oracle.fp32_eval("sin", np.array([0.0], np.float32))[0] == 0.0
    assert oracle.fp32_eval("cos", np.array([0.0], np.float32))[0] == 1.0
    assert oracle.fp32_eval("log", np.array([1.0], np.float32))[0] == 0.0
    assert oracle.fp32_eval("exp", np.array([0.0], np.float32))[0] == 1.0
    assert np.isneginf(oracle.fp32_eval("log", np.array([0.0], np.float32))[0])
    assert np.isnan(oracle.fp32_eval("asin", np.array([1.5], np.float32))[0])
