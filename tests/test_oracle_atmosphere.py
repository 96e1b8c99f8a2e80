"""Pins the oracle's atmosphere restatement (Atmosphere.slang, RayGen.slang:382-470) with closed forms: both trackers
estimate exp(-optical depth) of the Rayleigh + Mie + ozone profiles, which numpy integrates directly.  (No golden
data for the atmosphere exists in the reference tree: against the reference itself this part is parity unpinned.)"""
import numpy as np
import pytest

C_RAYLEIGH = np.array([5.802, 13.558, 33.100]) * 1e-6
C_MIE = (3.996 + 4.40) * 1e-6
C_OZONE = np.array([0.650, 1.881, 0.085]) * 1e-6


def optical_depth(atm, origin, direction, ch, n=400000):
    o = np.asarray(origin, np.float64); d = np.asarray(direction, np.float64)
    c = np.array(atm.planet_position[:], np.float64)
    R = atm.planet_radius + atm.atmosphere_height
    oc = o - c
    a, b, cc = d @ d, 2 * oc @ d, oc @ oc - R * R
    t1 = (-b + np.sqrt(b * b - 4 * a * cc)) / (2 * a)
    t0 = max((-b - np.sqrt(b * b - 4 * a * cc)) / (2 * a), 0.0)
    t = np.linspace(t0, t1, n)
    h = np.linalg.norm(o[None] + t[:, None] * d[None] - c[None], axis=1) - atm.planet_radius
    sigma = (C_RAYLEIGH[ch] * atm.rayleigh_multiplier[ch] * np.exp(-h / atm.rayleigh_density_falloff) +
             C_MIE * atm.mie_multiplier[ch] * np.exp(-h / atm.mie_density_falloff) +
             C_OZONE[ch] * atm.ozone_multiplier[ch] * np.exp(-np.abs(h - atm.ozone_peak) / atm.ozone_density_falloff))
    return float(np.trapezoid(sigma, t))


@pytest.mark.parametrize("direction,ch", [((0, -1, 0), 0), ((0, -1, 0), 2), ((0.6, -0.8, 0), 1), ((0.995, -0.0999, 0), 2), ((0, -0.05, -1), 0)])
def test_trackers_match_the_optical_depth_integral(vpt, oracle, direction, ch):
    atm = vpt.atmosphere()
    d = np.array(direction, np.float64); d /= np.linalg.norm(d)
    expect = np.exp(-optical_depth(atm, (0, 0, 0), d, ch))
    n = 40000
    tr, esc = oracle.atmosphere_estimators(atm, (0, 0, 0), d, ch, seed=7 + ch, n=n)
    sigma = np.sqrt(expect * (1 - expect) / n)
    assert abs(esc - expect) < 5 * sigma + 2e-3, (esc, expect)     # delta tracking: a Bernoulli estimator
    assert abs(tr - expect) < 5 * sigma + 2e-3, (tr, expect)       # ratio tracking + roulette: also 0/1-valued here


def test_planet_blocks_and_empty_atmosphere_is_clear(vpt, oracle):
    atm = vpt.atmosphere()
    tr, esc = oracle.atmosphere_estimators(atm, (0, 0, 0), (0, 1, 0), 1, n=200)        # straight down: the planet
    assert tr == 0.0
    clear = vpt.atmosphere(rayleigh_multiplier=(0, 0, 0), mie_multiplier=(0, 0, 0), ozone_multiplier=(0, 0, 0))
    assert oracle.atmosphere_estimators(clear, (0, 0, 0), (0, -1, 0), 0, n=50) == (1.0, 1.0)


def test_sky_colour_and_no_atmosphere_bits(vpt, oracle, scenes):
    """High sun: Rayleigh makes the sky blue (B > G > R); low sun: the horizon reddens.  Turning the atmosphere off
    restores the plain integrator bit for bit."""
    from test_oracle_volumes import empty_scene
    sc = empty_scene(vpt)
    sc.view_inverse = np.linalg.inv(vpt.scenes.look_at((0, 0, 0), (0, -0.3, -1), (0, 1, 0))).astype(np.float32)

    def sky(alt):
        o = oracle.Oracle(sc, 64, 36)
        o.set_params(vpt.default_params(max_depth=16, max_samples=1 << 30, sky_altitude=alt, sky_azimuth=40.0))
        o.set_atmosphere(vpt.atmosphere()); o.render(48)
        img = o.radiance(); o.close()
        return img[..., :3].reshape(-1, 3).mean(0), img
    noon, img = sky(-40.0)
    assert noon[2] > noon[1] > noon[0] > 0 and np.isfinite(img).all()
    dusk, img = sky(-3.0)
    assert img[-6:, :, 0].mean() > img[-6:, :, 2].mean()           # near the horizon: red over blue
    c = scenes("cornell_box")
    o = oracle.Oracle(c, 48, 27); P = vpt.default_params(max_depth=4); o.set_params(P); o.render(2); base = o.radiance()
    o.set_atmosphere(vpt.atmosphere()); o.render(2); lit = o.radiance()
    o.set_atmosphere(None); o.render(2); again = o.radiance(); o.close()
    assert np.array_equal(base, again) and not np.array_equal(base, lit)
