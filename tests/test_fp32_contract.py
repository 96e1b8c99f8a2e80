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


def test_acos_is_its_three_branch_definition_bit_for_bit(oracle):
    """acos_ evaluates asin once on a selected argument (vpt_fp32.h); its definition is three branches with an asin_ each.  Same bits on
    4 M random inputs, every branch boundary and the neighbours of +-0.5 and +-1 (an exhaustive run over all 2^32 inputs was done once:
    0 differences)."""
    rng = np.random.RandomState(11)
    x = np.concatenate([(rng.rand(1 << 22) * 2.0 - 1.0).astype(np.float32),
                        np.array([0.0, -0.0, 0.5, -0.5, 1.0, -1.0], np.float32)])
    for v in (0.5, -0.5, 1.0, -1.0):
        c = np.float32(v)
        x = np.concatenate([x, np.array([np.nextafter(c, np.float32(2)), np.nextafter(c, np.float32(-2))], np.float32)])
    x = x[np.abs(x) <= 1.0]
    f32 = np.float32
    a_lo = oracle.fp32_eval("asin", np.sqrt(f32(0.5) * (f32(1.0) + x)))      # float32 numpy arithmetic = the IEEE operations of the C side
    a_hi = oracle.fp32_eval("asin", np.sqrt(f32(0.5) * (f32(1.0) - x)))
    a_mid = oracle.fp32_eval("asin", x)
    ref = np.where(x < f32(-0.5), f32(3.14159265358979323846) - f32(2.0) * a_lo, np.where(x > f32(0.5), f32(2.0) * a_hi, f32(1.57079632679489661923) - a_mid)).astype(np.float32)
    got = oracle.fp32_eval("acos", x)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


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


# ---- the leaf primitives BOTH sides compile (vpt_fp32.h), against independent float64 / numpy restatements --------------
# HIP-vs-oracle parity is blind to a bug in shared code; these are the only checks that can see one.

def _mt64(o, d, v0, e1, e2):
    """Moeller-Trumbore in float64: returns det, t, u, v."""
    p = np.cross(d, e2)
    det = (e1 * p).sum(1)
    inv = 1.0 / det
    s = o - v0
    u = (s * p).sum(1) * inv
    q = np.cross(s, e1)
    v = (d * q).sum(1) * inv
    t = (e2 * q).sum(1) * inv
    return det, t, u, v


def test_ray_triangle_against_float64(oracle):
    rng = np.random.default_rng(21)
    n = 400000
    v0 = rng.uniform(-5, 5, (n, 3)); e1 = rng.uniform(-2, 2, (n, 3)); e2 = rng.uniform(-2, 2, (n, 3))
    # aim at a point of the triangle's plane near the triangle, from a random origin: about half of the rays hit
    a, b = rng.uniform(-0.3, 1.3, n), rng.uniform(-0.3, 1.3, n)
    target = v0 + e1 * a[:, None] + e2 * b[:, None]
    o = target + rng.normal(size=(n, 3)) * rng.uniform(0.5, 8, (n, 1))
    d = target - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
    rows = np.concatenate([o, d, v0, e1, e2, np.full((n, 1), 1e-4), np.full((n, 1), 1e6)], 1).astype(np.float32)
    got = oracle.leaf_eval("ray_triangle", rows)
    r64 = rows.astype(np.float64)
    det, t, u, v = _mt64(r64[:, 0:3], r64[:, 3:6], r64[:, 6:9], r64[:, 9:12], r64[:, 12:15])
    hit64 = (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 1e-4) & (t < 1e6)
    # conditioning: the relative error of u, v, t grows like eps / |cos(angle between ray and plane)|; keep well-conditioned rays
    scale = np.linalg.norm(r64[:, 9:12], axis=1) * np.linalg.norm(r64[:, 12:15], axis=1)
    well = np.abs(det) > 0.05 * scale
    margin = np.minimum.reduce([u, 1 - u, v, 1 - u - v, t - 1e-4])
    clear = well & (np.abs(margin) > 1e-4)            # decisions are only compared away from the edges
    assert clear.mean() > 0.5 and hit64[clear].mean() > 0.15
    assert np.array_equal(got[clear, 0] > 0.5, hit64[clear])
    h = clear & hit64
    # with |o - v0| up to ~30 and |det| >= 5 % of |e1||e2| the fp32 evaluation is good to a few 1e-5 absolute in u, v and
    # relative in t (cancellation in o - v0 dominates)
    assert np.abs(got[h, 2] - u[h]).max() < 1e-4 and np.abs(got[h, 3] - v[h]).max() < 1e-4
    assert (np.abs(got[h, 1] - t[h]) / np.maximum(np.abs(t[h]), 1.0)).max() < 1e-4
    # grazing rays (1e-3 rad to the plane): a clear float64 hit / miss inside the triangle's interior is still decided the same way
    d2 = e1 * rng.uniform(-1, 1, (n, 1)) + e2 * rng.uniform(-1, 1, (n, 1)) + np.cross(e1, e2) * 1e-3
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    o2 = target - d2 * rng.uniform(1, 10, (n, 1))
    rows2 = np.concatenate([o2, d2, v0, e1, e2, np.full((n, 1), 1e-4), np.full((n, 1), 1e6)], 1).astype(np.float32)
    got2 = oracle.leaf_eval("ray_triangle", rows2)
    q = rows2.astype(np.float64)
    det, t, u, v = _mt64(q[:, 0:3], q[:, 3:6], q[:, 6:9], q[:, 9:12], q[:, 12:15])
    hit64 = (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > 1e-4)
    margin = np.minimum.reduce([u, 1 - u, v, 1 - u - v])
    clear = (np.abs(margin) > 0.05) & (np.abs(det) > 1e-5 * np.linalg.norm(q[:, 9:12], axis=1) * np.linalg.norm(q[:, 12:15], axis=1))
    assert clear.sum() > 10000 and np.array_equal(got2[clear, 0] > 0.5, hit64[clear])
    # det == 0 (ray in the plane) never hits; a miss leaves the decision false whatever t, u, v hold
    flat = np.array([[0, 0, 1, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 0, 1e-4, 1e6]], np.float32)   # triangle in the plane z = 0, ray along +x at z = 1
    assert oracle.leaf_eval("ray_triangle", flat)[0, 0] == 0.0


def test_texel_coords_against_numpy(oracle):
    rng = np.random.default_rng(22)
    for size in (1, 2, 7, 64, 1024):
        u = np.concatenate([rng.uniform(-3, 3, 20000), np.array([0.0, 1.0, -1.0, 0.5 / size, 1 - 0.5 / size, -0.5 / size, 1e12, -1e12, np.nan])]).astype(np.float32)
        for repeat in (1.0, 0.0):
            rows = np.stack([u, np.full_like(u, size), np.full_like(u, repeat)], 1)
            got = oracle.leaf_eval("texel_coords", rows)
            x = u.astype(np.float32) * np.float32(size) - np.float32(0.5)          # Vulkan: unnormalised coordinate - 0.5 (fp32 as specified)
            x = np.where(np.abs(x) < 1e9, x, np.float32(0.0)).astype(np.float32)   # the contract's guard for huge / NaN coordinates
            fl = np.floor(x.astype(np.float64))
            w = (x.astype(np.float64) - fl)
            i0, i1 = fl.astype(np.int64), fl.astype(np.int64) + 1
            if repeat:
                i0, i1 = np.mod(i0, size), np.mod(i1, size)                        # REPEAT (PathTracer.cpp:84-91)
            else:
                i0, i1 = np.clip(i0, 0, size - 1), np.clip(i1, 0, size - 1)        # CLAMP_TO_EDGE (LUT sampler, PathTracer.cpp:93-94)
            assert np.array_equal(got[:, 0].astype(np.int64), i0) and np.array_equal(got[:, 1].astype(np.int64), i1), (size, repeat)
            assert np.abs(got[:, 2] - w).max() < 1e-6 and (got[:, 2] >= 0).all() and (got[:, 2] < 1.0 + 1e-7).all()


def test_unorm8_texel_decode_equals_the_division_for_every_byte(oracle):
    """The HIP texel fetch decodes a byte as b * r + residual step instead of b / 255.0f (shading.hpp tex_fetch); the oracle divides.
    All 256 inputs, against numpy's correctly rounded float32 division and against the float64 quotient rounded once."""
    b = np.arange(256, dtype=np.float32)
    got = oracle.leaf_eval("unorm8_to_float", b)[:, 0]
    assert np.array_equal(got, b / np.float32(255.0))
    assert np.array_equal(got, (b.astype(np.float64) / 255.0).astype(np.float32))


def test_lut_layer_rounds_to_nearest_even(oracle):
    layers = 32
    x = np.concatenate([np.arange(-2, 34, 0.25), np.array([0.5, 1.5, 2.5, 30.5, 31.5, np.nan, 1e9, -1e9])]).astype(np.float32)
    got = oracle.leaf_eval("lut_layer", np.stack([x, np.full_like(x, layers)], 1))[:, 0]
    ref = np.rint(np.clip(np.nan_to_num(x.astype(np.float64), nan=0.0), 0, layers - 1))   # numpy rint is round-half-to-even
    assert np.array_equal(got, ref.astype(np.float32))
    assert got[list(x).index(np.float32(0.5))] == 0 and got[list(x).index(np.float32(1.5))] == 2 and got[list(x).index(np.float32(2.5))] == 2


def test_refract_reflect_normalize_smoothstep_unorm8_against_float64(oracle):
    rng = np.random.default_rng(23)
    n = 100000
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    inc = rng.normal(size=(n, 3)); inc /= np.linalg.norm(inc, axis=1, keepdims=True)
    inc[(inc * nrm).sum(1) > 0] *= -1                                        # incident against the normal, as the shaders call it
    eta = rng.choice([1 / 1.5, 1.5, 1 / 1.0001, 1.33], n)
    rows = np.concatenate([inc, nrm, eta[:, None]], 1).astype(np.float32)
    got = oracle.leaf_eval("refract", rows).astype(np.float64)
    q = rows.astype(np.float64)
    ni = (q[:, 3:6] * q[:, 0:3]).sum(1)
    k = 1 - q[:, 6] ** 2 * (1 - ni * ni)
    ref = np.where((k >= 0)[:, None], q[:, 0:3] * q[:, 6:7] - q[:, 3:6] * (q[:, 6] * ni + np.sqrt(np.maximum(k, 0)))[:, None], 0.0)   # GLSL refract
    clear = np.abs(k) > 1e-3                                                 # sqrt(k) amplifies rounding near the critical angle
    assert np.abs(got[clear] - ref[clear]).max() < 1e-5                       # eps / (2 sqrt(k)) at k = 1e-3, times a few operations
    assert (got[k < -1e-3] == 0).all() and (np.abs(got[k > 1e-3]).sum(1) > 0).all()          # total internal reflection returns the zero vector
    got = oracle.leaf_eval("reflect", rows[:, :6]).astype(np.float64)
    assert np.abs(got - (q[:, 0:3] - 2 * ni[:, None] * q[:, 3:6])).max() < 1e-6
    v = (rng.normal(size=(n, 3)) * rng.uniform(1e-3, 1e3, (n, 1))).astype(np.float32)
    got = oracle.leaf_eval("normalize", v).astype(np.float64)
    assert np.abs(got - v.astype(np.float64) / np.linalg.norm(v.astype(np.float64), axis=1, keepdims=True)).max() < 3e-7
    e = rng.uniform(-2, 2, (n, 3)).astype(np.float32); e[:, 1] = e[:, 0] + np.abs(e[:, 1]) + np.float32(0.01)
    got = oracle.leaf_eval("smoothstep", e)[:, 0].astype(np.float64)
    t = np.clip((e[:, 2].astype(np.float64) - e[:, 0]) / (e[:, 1].astype(np.float64) - e[:, 0]), 0, 1)
    assert np.abs(got - t * t * (3 - 2 * t)).max() < 2e-6
    c = np.concatenate([rng.uniform(-0.5, 1.5, n), np.array([0.0, 1.0, 0.5 / 255, 1.5 / 255, 2.5 / 255, np.nan])]).astype(np.float32)
    got = oracle.leaf_eval("unorm8", c)[:, 0]
    ref = np.rint(np.clip(np.nan_to_num(c, nan=0.0), 0, 1).astype(np.float32) * np.float32(255.0))      # RNE(saturate(c) * 255), NaN -> 0
    assert np.array_equal(got, ref)


def test_triangle_degenerate_and_hit_is_local(oracle):
    rng = np.random.default_rng(24)
    n = 50000
    e1 = rng.uniform(-2, 2, (n, 3)); e2 = rng.uniform(-2, 2, (n, 3))
    got = oracle.leaf_eval("triangle_degenerate", np.concatenate([e1, e2], 1))[:, 0]
    sin2 = (np.cross(e1, e2) ** 2).sum(1) / ((e1 ** 2).sum(1) * (e2 ** 2).sum(1))
    assert not got[sin2 > 1e-9].any()                                         # proper triangles are kept
    e2s = e1 * rng.uniform(-3, 3, (n, 1))                                     # exactly parallel edges
    assert oracle.leaf_eval("triangle_degenerate", np.concatenate([e1, e2s], 1))[:, 0].all()
    assert oracle.leaf_eval("triangle_degenerate", np.concatenate([e1, np.zeros_like(e1)], 1))[:, 0].all()
    # hit_is_local: the true hit point of a ray through the triangle's interior lies in the triangle's box; a t far outside does not
    v0 = rng.uniform(-5, 5, (n, 3)); a, b = rng.uniform(0.2, 0.4, n), rng.uniform(0.2, 0.4, n)
    target = v0 + e1 * a[:, None] + e2 * b[:, None]
    o = target + rng.normal(size=(n, 3)) * 5
    t = np.linalg.norm(target - o, axis=1); d = (target - o) / t[:, None]
    ok = sin2 > 1e-3
    rows = np.concatenate([o, d, v0, e1, e2, t[:, None]], 1)
    assert oracle.leaf_eval("hit_is_local", rows)[ok, 0].all()
    ext = np.linalg.norm(e1, axis=1) + np.linalg.norm(e2, axis=1)
    rows[:, 15] = t + 2 * ext + 1.0
    assert not oracle.leaf_eval("hit_is_local", rows)[ok, 0].any()
