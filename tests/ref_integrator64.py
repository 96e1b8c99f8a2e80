"""A second, independent restatement of the reference's per-sample integrator — float64 Python, written from the Slang sources
(RayGen.slang:9-160, ClosestHit.slang:20-378, Miss.slang:8-77, Surface.slang:26-147, Sampler.slang:286-422, RTCommon.slang:47-64,
124-136, Material.slang via tests/test_oracle_bsdf_fp64.py) and NOT from oracle/oracle.cpp.  PathTracer.cpp:1161-1296 for the environment tables).  Scope: any textures (LINEAR / REPEAT, mip 0) and environment map, the medium inside a glass mesh (Beer's law, Henyey-Greenstein walk),
homogeneous box volumes with the Henyey-Greenstein phase function (RayGen.slang:162-372, Volume.slang:190-223, 256-286, 358-443);
the atmosphere (Atmosphere.slang, RayGen.slang:214-250, 382-470, Sampler.slang:430-463; its sphere intersections and heights in
float32, as the shader computes them: at a planet radius of 6.36e6 they are cancellation-limited and float64 would be a different
function); density grids as the backend takes them (dense, include/vpt.h), their block walk in float32 (see hetero_walk); ray-query shadow tests; brute-force intersection.  Used by
tests/test_oracle_integrator_fp64.py to hold the oracle's per-sample values (orc_pixel_samples) against it.

Test infrastructure only."""
import numpy as np

from test_oracle_bsdf_fp64 import Mat64, Rng64, _pcg, _norm, sample64

MAX_DEPTH = 1000000   # Defines.slang:17
FLAG_SKY_MIS, FLAG_MESH_MIS, FLAG_SHOW_ENV_DIRECTLY, FLAG_GEOMETRY_NORMALS, FLAG_ENERGY_COMPENSATION, FLAG_FURNACE = 1, 2, 4, 8, 16, 32   # the reference's #defines as vpt_params.flags bits
FLAG_RAY_QUERIES = 64


def shadow_hit(S, P, o, d):
    """DoesRayIntersectWithAS (RTCommon.slang:47-84) -> (closest hit or None, are payload.TriangleIdx / InstanceIdx defined).
    USE_RAY_QUERIES: the direction as it is, [1e-4, 1e6], the committed hit's identity.  Otherwise the TraceRay form (:64-84, MissShadow.slang:4-9):
    normalised direction, [1e-5, 1000], accept-first-hit with the closest-hit shader skipped — nothing writes the identity words, and a compare
    against them is pinned as 'never equal' (include/vpt.h VPT_FLAG_RAY_QUERIES)."""
    if P.flags & FLAG_RAY_QUERIES:
        return S.closest(o, d, 0.0001, 1000000.0), True
    return S.closest(o, _norm(d), 0.00001, 1000.0), False


def power_heuristics(a, b):   # RTCommon.slang:124-127
    with np.errstate(all="ignore"):
        return np.float64(a) ** 2 / (np.float64(a) ** 2 + np.float64(b) ** 2)


class Scene64:
    def __init__(self, sc, width, height):
        self.W, self.H = width, height
        self.view_inv = np.asarray(sc.view_inverse, np.float64)
        self.proj_inv = np.asarray(sc.projection_inverse(width / height), np.float64)
        self.materials = sc.materials
        self.inst = []           # (mesh, material, M 4x4, inverse 3x3)
        v0, e1, e2, ids = [], [], [], []
        for ii, (mesh, mat, xf) in enumerate(sc.instances):
            M = np.asarray(xf, np.float64)
            self.inst.append((mesh, mat, M, np.linalg.inv(M[:3, :3])))
            vert, idx = sc.meshes[mesh]
            P = vert["position"].astype(np.float64) @ M[:3, :3].T + M[:3, 3]
            t = idx.reshape(-1, 3)
            for k, (a, b, c) in enumerate(t):
                v0.append(P[a]); e1.append(P[b] - P[a]); e2.append(P[c] - P[a]); ids.append((ii, k))
        self.meshes = sc.meshes
        self.v0, self.e1, self.e2 = np.array(v0), np.array(e1), np.array(e2)
        self.ids = ids
        # emissive-mesh list: instances whose material emits, in instance order (PathTracer.cpp:449-469)
        self.emissive = [i for i, (me, ma, _, _) in enumerate(self.inst) if any(c != 0 for c in sc.materials[ma]["emissive_color"])]
        self.build_env(np.asarray(sc.env, np.float32))
        self.textures = [np.asarray(t) for t in sc.textures]
        self.volumes = []   # set_volumes(): dicts with corner_min, corner_max, color, emissive_color, density, anisotropy
        self.atm = None     # set_atmosphere(): a vpt_atmosphere
        self.phase = 0      # 0 Henyey-Greenstein, 1 Draine, 2 Henyey-Greenstein + Draine (the reference's PHASE_FUNCTION_* defines)

    def set_atmosphere(self, a):
        self.atm = a

    def set_volumes(self, vols, grids=()):
        """grids: the dense density grids (float32 [z, y, x] raw densities, as handed to vpt_add_density_grid) a volume's density_data_index
        refers to.  What AddDensityDataToVolume derives from a grid (PathTracer.cpp:1390-1442) is derived here the same way: the maximum, and
        the 32 x 32 x 32 table of per-block maxima of density / max with y flipped ("Y has to be flipped for vulkan", :1435)."""
        self.volumes = [dict(lo=np.array(v.corner_min[:], np.float64), hi=np.array(v.corner_max[:], np.float64), color=np.array(v.color[:], np.float64),
                             emissive=np.array(v.emissive_color[:], np.float64), density=float(v.density), g=float(v.anisotropy), alpha=float(v.alpha),
                             droplet=float(v.droplet_size), grid=None, sharpness=float(v.grid_sharpness)) for v in vols]
        for v, d in zip(vols, self.volumes):
            assert not v.approximated_scattering and not v.has_temperature_data, "no approximated scattering, no temperature grids"
            if v.density_data_index >= 0:
                g = np.asarray(grids[v.density_data_index], np.float32)
                dz, dy, dx = g.shape
                mx = float(g.max())
                norm = np.clip(g.astype(np.float64) / mx, 0.0, 1.0)[:, ::-1, :]          # y flipped
                bx = (np.arange(dx) * 32) // dx; by = (np.arange(dy) * 32) // dy; bz = (np.arange(dz) * 32) // dz
                bm = np.zeros((32, 32, 32))                                               # [bz, by, bx]
                np.maximum.at(bm, (bz[:, None, None], by[None, :, None], bx[None, None, :]), norm)
                d["grid"] = dict(values=g, max=mx, block_max=bm, dim=(dx, dy, dz))

    def tex(self, ti, uv):
        """uTextures[ti].SampleLevel(uTextureSampler, uv, 0): UNORM8, bilinear, REPEAT (PathTracer.cpp:84-91) -> rgba."""
        t = self.textures[ti]
        h, w, c = t.shape
        def axis(x, n):
            x = x * n - 0.5
            fl = np.floor(x)
            return int(fl) % n, (int(fl) + 1) % n, x - fl
        x0, x1, fx = axis(uv[0], w); y0, y1, fy = axis(uv[1], h)
        def texel(y, x):
            p = t[y, x].astype(np.float64) / 255.0
            return p if c == 4 else np.array([p[0], 0.0, 0.0, 1.0])
        a = texel(y0, x0) + (texel(y0, x1) - texel(y0, x0)) * fx
        b = texel(y1, x0) + (texel(y1, x1) - texel(y1, x0)) * fx
        return a + (b - a) * fy

    def build_env(self, env):
        """LoadEnvironmentMap (PathTracer.cpp:1161-1296): importance = solid angle x max(r, g, b), the alias map with its partition
        quirk (the low-energy list is filled from index 1), the pdf in alpha.  Sequential float32 arithmetic, as the C++ does it."""
        f = np.float32
        h, w = env.shape[:2]
        px = env.reshape(-1, 4).copy()
        size = w * h
        imp = np.zeros(size, f)
        cos0 = f(1.0)
        step_phi = f(2.0) * f(np.pi) / f(w); step_theta = f(np.pi) / f(h)
        mx = np.maximum(px[:, 0], np.maximum(px[:, 1], px[:, 2]))
        for y in range(h):
            cos1 = f(np.cos(f(y + 1) * step_theta))
            area = f(f(cos0 - cos1) * step_phi)
            cos0 = cos1
            imp[y * w:(y + 1) * w] = area * mx[y * w:(y + 1) * w]
        total = f(0.0)
        for v in imp:
            total = f(total + v)                       # std::accumulate, in order
        avg = f(total / f(size))
        importance = np.zeros(size, f) if avg == 0 else (imp / avg).astype(f)
        alias = np.arange(size, dtype=np.int64)
        part = np.zeros(size + 1, np.int64)
        lo, hi = 0, size
        for i in range(size):
            if importance[i] < f(1.0):
                lo += 1
                if lo < size:
                    part[lo] = i
            else:
                hi -= 1; part[hi] = i
        lo = 0
        while lo < hi and hi < size:
            li, hh = part[lo], part[hi]
            alias[li] = hh
            importance[hh] = f(importance[hh] - f(f(1.0) - importance[li]))
            if importance[hh] < f(1.0):
                hi += 1
            lo += 1
        px[:, 3] = 0.0 if total == 0 else (mx / total).astype(f)
        self.env_w, self.env_h = w, h
        self.env_px = px.reshape(h, w, 4).astype(np.float64)
        self.alias, self.importance = alias, importance.astype(np.float64)
        self.env_black = bool(np.abs(px).max() == 0.0)

    def env_lookup(self, u, v):
        """uEnvMapTexture.SampleLevel: bilinear, REPEAT, texel centres at +0.5."""
        def axis(c, n):
            x = c * n - 0.5
            fl = np.floor(x)
            return int(fl) % n, (int(fl) + 1) % n, x - fl
        x0, x1, fx = axis(u, self.env_w); y0, y1, fy = axis(v, self.env_h)
        e = self.env_px
        a = e[y0, x0] + (e[y0, x1] - e[y0, x0]) * fx
        b = e[y1, x0] + (e[y1, x1] - e[y1, x0]) * fx
        return a + (b - a) * fy

    def closest(self, o, d, tmin, tmax):
        """Two-sided Moller-Trumbore over every triangle; ties in t -> smaller global id.  -> (t, u, v, gid) or None."""
        p = np.cross(d, self.e2)
        det = (self.e1 * p).sum(1)
        with np.errstate(all="ignore"):
            inv = 1.0 / det
            s = o - self.v0
            u = (s * p).sum(1) * inv
            q = np.cross(s, self.e1)
            v = (q * d).sum(1) * inv
            t = (self.e2 * q).sum(1) * inv
        ok = (det != 0) & (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > tmin) & (t < tmax)
        if not ok.any():
            return None
        tt = np.where(ok, t, np.inf)
        g = int(np.argmin(tt))   # argmin returns the first minimum = the smaller id on exact ties
        return float(t[g]), float(u[g]), float(v[g]), g


def sample_value(S, luts, x, y, frame, P):
    """accumulatedLight / SampleCount of pixel (x, y) in dispatch `frame` (RayGen.slang:27-130): one sampler per pixel and dispatch,
    SampleCount samples drawn from it one after the other."""
    rng = Rng64((y + S.W * x + _pcg((P.base_seed + frame) & 0xffffffff)) & 0xffffffff)
    acc = np.zeros(3)
    for _ in range(P.samples_per_frame):
        acc = acc + one_sample(S, luts, x, y, rng, P)
    return acc / float(P.samples_per_frame)


def one_sample(S, luts, x, y, rng, P):
    j0 = rng.uf() * 1.0 - 0.5; j1 = rng.uf() * 1.0 - 0.5
    d2 = np.array([(x + 0.5 + j0) / S.W, (y + 0.5 + j1) / S.H]) * 2.0 - 1.0
    origin = (S.view_inv @ np.array([0.0, 0.0, 0.0, 1.0]))[:3]
    target = (S.proj_inv @ np.array([d2[0], d2[1], 1.0, 1.0]))[:3]
    direction = (S.view_inv @ np.append(_norm(target), 0.0))[:3]
    focus = origin + direction * max(P.focus_distance, 0.001)
    u1, u2 = rng.uf(), rng.uf()                       # RandomCircleVec: drawn whatever the strength is
    off = np.array([np.sqrt(u2) * np.cos(2 * np.pi * u1), np.sqrt(u2) * np.sin(2 * np.pi * u1)]) * 0.5 * P.dof_strength
    origin = origin + off[0] * S.view_inv[:3, 0] + off[1] * S.view_inv[:3, 1]
    direction = _norm(focus - origin)
    pay = dict(depth=0, origin=origin, direction=direction, bxdf=np.ones(3), pdf=1.0, emitted=np.zeros(3), in_medium=False,
               med_density=0.0, med_aniso=0.0, med_color=np.zeros(3), vdepth=0, cchan=-1)
    thr, light = np.ones(3), np.zeros(3)
    while pay["depth"] < P.max_depth:
        rd = _norm(pay["direction"])
        pay["emitted"] = np.zeros(3)
        if S.atm is not None and atm_height(S.atm, pay["origin"]) < 0.0:
            break                                                         # below the planet's surface (RayGen.slang:76-84)
        if (S.volumes or S.atm is not None) and scattered_in_volume(S, pay, rng, P):
            pass
        else:
            hit = S.closest(pay["origin"], rd, 0.01, 100000.0)
            if hit is None:
                miss(S, pay, P)
            else:
                closest_hit(S, luts, pay, rd, hit, rng, P)
        contrib = pay["emitted"] * thr
        if pay["depth"] != 1:
            lum = float(np.dot(contrib, [0.212671, 0.715160, 0.072169]))
            contrib = contrib * (P.max_luminance / max(lum, P.max_luminance))
        light = light + contrib
        with np.errstate(all="ignore"):
            thr = thr * (pay["bxdf"] / pay["pdf"])
        p = min(max(thr[0], max(thr[1], thr[2])), 1.0) if not np.isnan(thr).any() else float("nan")
        if p < rng.uf():
            break
        with np.errstate(all="ignore"):
            thr = thr / p
    if not np.isfinite(light).all():
        return np.zeros(3)
    if pay["cchan"] == -1:
        return light
    out = np.zeros(3); out[pay["cchan"]] = light[pay["cchan"]]           # a split path carries one colour channel (RayGen.slang:118-128)
    return out


def sample_hg(d, g, rng):   # Sampler.slang:168-192
    r0, r1 = rng.uf(), rng.uf()
    if abs(g) < 1e-5:
        ct = 2.0 * r0 - 1.0
    else:
        sq = (1.0 - g * g) / (1.0 - g + 2.0 * g * r0)
        ct = (1.0 + g * g - sq * sq) / (2.0 * g)
    phi = 2.0 * np.pi * r1
    st = np.sqrt(1.0 - ct * ct)
    nd = np.array([st * np.cos(phi), st * np.sin(phi), ct])
    up = np.array([0.0, 1.0, 0.0]) if abs(d[1]) < 0.9999999 else np.array([0.0, 0.0, 1.0])
    t = _norm(np.cross(up, d)); b = np.cross(d, t)
    return _norm(nd[0] * t + nd[1] * b + nd[2] * d)


def box_hit(o, d, lo, hi):   # Volume.slang:190-213 (as written, quirks included: the x slab appears twice in each max / min)
    with np.errstate(all="ignore"):
        inv = 1.0 / d
        t0, t1 = (lo - o) * inv, (hi - o) * inv
    sm, bg = np.minimum(t0, t1), np.maximum(t0, t1)
    tmin = max(max(sm[0], sm[1]), max(sm[0], sm[2])); tmax = min(min(bg[0], bg[1]), min(bg[0], bg[2]))
    if tmax < 0.0 or tmin > tmax:
        return -1.0, -1.0
    return tmin, tmax


def phase_hg(V, L, g):   # RTCommon.slang:213-220
    if g == 0.0:
        return 1.0 / (4.0 * np.pi)
    return (1.0 / (4.0 * np.pi)) * ((1.0 - g * g) / (1.0 + g * g - 2.0 * g * float(np.dot(V, L))) ** 1.5)


def phase_draine(V, L, g, a):   # RTCommon.slang:222-227
    c = float(np.dot(V, L))
    return ((1 - g * g) * (1 + a * c * c)) / (4.0 * (1 + (a * (1 + 2 * g * g)) / 3.0) * np.pi * (1 + g * g - 2 * g * c) ** 1.5)


def hgd_params(d):   # Volume.slang:396-407 == Sampler.slang:268-272
    return (np.exp(-(0.0990567 / (d - 1.67154))), np.exp(-(2.20679 / (d + 3.91029)) - 0.428934), np.exp(3.62489 - (8.29288 / (d + 5.52825))),
            np.exp(-(0.599085 / (d - 0.641583)) - 0.665888))


def eval_phase(S, v, V, L):   # Volume.slang:377-407 (no approximated scattering: the anisotropy does not change with depth)
    if S.phase == 0:
        return phase_hg(V, L, v["g"])
    if S.phase == 1:
        return phase_draine(V, L, v["g"], v["alpha"])
    ghg, gd, ad, wd = hgd_params(v["droplet"])
    hg, dr = phase_hg(V, L, ghg), phase_draine(V, L, gd, ad)
    return hg + (dr - hg) * wd


def to_world(d, nd):
    up = np.array([0.0, 1.0, 0.0]) if abs(d[1]) < 0.9999999 else np.array([0.0, 0.0, 1.0])
    t = _norm(np.cross(up, d)); b = np.cross(d, t)
    return _norm(nd[0] * t + nd[1] * b + nd[2] * d)


def sample_draine(d, g, a, rng):   # Sampler.slang:216-264; the closed-form root in float32, term for term (it cancels catastrophically)
    r0, r1 = rng.uf(), rng.uf()
    if abs(g) < 1e-5:
        ct = 2.0 * r0 - 1.0
    elif abs(a) < 1e-5:
        sq = (1.0 - g * g) / (1.0 - g + 2.0 * g * r0)
        ct = (1.0 + g * g - sq * sq) / (2.0 * g)
    else:
        f = np.float32
        g, a, x = f(g), f(a), f(r0)
        g2 = g * g; g3 = g * g2; g4 = g2 * g2; g6 = g2 * g4
        p2 = (1 + g2) * (1 + g2)
        T1a = -a + a * g4
        T1a3 = T1a * T1a * T1a
        T2 = -1296 * (-1 + g2) * (a - a * g2) * (T1a) * (4 * g2 + a * p2)
        T3 = 3 * g2 * (1 + g * (-1 + 2 * x)) + a * (2 + g2 + g3 * (1 + 2 * g2) * (-1 + 2 * x))
        T4a = 432 * T1a3 + T2 + 432 * (a - a * g2) * T3 * T3
        T4b = -144 * a * g2 + 288 * a * g4 - 144 * a * g6
        T4b3 = T4b * T4b * T4b
        with np.errstate(all="ignore"):
            T4 = T4a + np.sqrt(-4 * T4b3 + T4a * T4a)
            T4p3 = f(np.power(T4, f(1.0 / 3.0)))
            c2 = f(np.power(f(2.0), f(1.0 / 3.0)))
            T6 = (2 * T1a + (48 * c2 * (-(a * g2) + 2 * a * g4 - a * g6)) / T4p3 + T4p3 / (f(3.0) * c2)) / (a - a * g2)
            T5 = 6 * (1 + g2) + T6
            inner = f(-0.5) * np.sqrt(T5) + np.sqrt(6 * (1 + g2) - (8 * T3) / (a * (-1 + g2) * np.sqrt(T5)) - T6) / f(2.0)
            ct = float((1 + g2 - inner * inner) / (f(2.0) * g))
    phi = 2.0 * np.pi * r1
    st = np.sqrt(1.0 - ct * ct)
    return to_world(d, np.array([st * np.cos(phi), st * np.sin(phi), ct]))


def sample_phase(S, v, d, vdepth, rng):   # Volume.slang:358-375
    if S.phase == 0:
        return sample_hg(d, v["g"], rng)
    if S.phase == 1:
        return sample_draine(d, v["g"], v["alpha"], rng)
    ghg, gd, ad, wd = hgd_params(v["droplet"])
    ghg = max(ghg, 0.0) ** (1.0 + vdepth); gd = max(gd, 0.0) ** (1.0 + vdepth)     # sampling narrows with depth, evaluation does not (upstream)
    return sample_hg(d, ghg, rng) if rng.uf() < wd else sample_draine(d, gd, ad, rng)


def sample_grid(v, rng, x):
    """SampleNanoVDBBuffer (Volume.slang:69-117) on the dense grid the backend takes in place of the NanoVDB tree (include/vpt.h
    vpt_add_density_grid): position normalised in the box, y flipped, scaled to the grid, floor to a voxel, +-1 voxel of jitter per axis
    (three raw PCG draws, `% 3 - 1`), clamp to the grid; value / max * GridSharpness clamped to [0, 1]."""
    g = v["grid"]
    n = (x - v["lo"]) / (v["hi"] - v["lo"])
    n[1] = 1.0 - n[1]
    dx, dy, dz = g["dim"]
    c = [int(np.floor(n[0] * dx)), int(np.floor(n[1] * dy)), int(np.floor(n[2] * dz))]
    for a in range(3):
        c[a] += int(rng.raw() % 3) - 1
    cx, cy, cz = min(max(c[0], 0), dx - 1), min(max(c[1], 0), dy - 1), min(max(c[2], 0), dz - 1)
    return min(max(float(g["values"][cz, cy, cx]) / g["max"] * v["sharpness"], 0.0), 1.0)


_f = np.float32


def box_hit32(o, d, lo, hi):   # ComputeRayAABBIntersection (Volume.slang:190-213) in float32, term for term
    with np.errstate(all="ignore"):
        inv = _f(1.0) / d
        t0, t1 = (lo - o) * inv, (hi - o) * inv
    sm, bg = np.minimum(t0, t1), np.maximum(t0, t1)
    tmin = max(max(sm[0], sm[1]), max(sm[0], sm[2])); tmax = min(min(bg[0], bg[1]), min(bg[0], bg[2]))
    if tmax < 0.0 or tmin > tmax:
        return _f(-1.0), _f(-1.0)
    return tmin, tmax


def hetero_walk(S, v, o, d, rng, near, far, transmittance):
    """ProcessHeterogeneousVolumeScattering (Volume.slang:299-348: delta tracking, returns the scatter distance or -1) and
    ProcessHeterogeneousVolumeTransmittance (:448-520: ratio tracking with roulette, returns the transmittance): one loop, block by
    block through the 32^3 table of majorants; the two differ in what a real collision does and in the step limit.
    In FLOAT32, term for term: every block crossing re-enters the volume `epsilon` (1e-4 of the box) beyond the exit point, positions
    are formed as origin + direction * (tEnter + t + epsilon) at distances of 10-20 units, so about one crossing in a few thousand
    lands on the other side of a block face than float64 would put it — one more or one less iteration, i.e. one more or one less
    random draw, which changes the whole remainder of the sample (measured with this function in float64: 4 % of the samples of a
    depth-1 render differ, 19 % at depth 8).  Float64 is a different function here, as with the atmosphere's sphere intersections."""
    o, d, lo, hi = o.astype(_f), d.astype(_f), v["lo"].astype(_f), v["hi"].astype(_f)
    eps = _f(0.0001) * _f((hi - lo).max())
    t_enter, t_exit = _f(max(near, 0.0)), _f(far)
    bs = (hi - lo) / _f(32.0)

    def block_info(pos):   # CalculateBlockInfo, Volume.slang:129-147
        rel = (pos - lo) / (hi - lo)
        idx = np.clip(np.trunc(rel * _f(32.0)).astype(np.int64), 0, 31)                   # int3(...) truncates
        blo = lo + bs * idx.astype(_f)
        return _f(v["grid"]["block_max"][idx[2], idx[1], idx[0]]), blo, blo + bs
    bmax, blo, bhi = block_info(o + d * (t_enter + eps))
    t, T = _f(0.0), _f(1.0)
    for _ in range(1000 if transmittance else 10000):
        cur = o + d * (t_enter + t + eps)
        bn, bf = box_hit32(cur, d, blo, bhi)
        maxd = bmax * _f(v["density"])
        with np.errstate(all="ignore"):
            sampled = _f(-np.log(_f(rng.uf()))) / maxd   # SampleScatteringDistance: -log(u) / density (a zero majorant gives +inf: on to the next block)
        if bf <= 0.0:                                   # "ray doesn't intersect block properly": advance by epsilon
            t = t + eps
            if t_enter + t > t_exit:
                break
            bmax, blo, bhi = block_info(o + d * (t_enter + t + eps))
            continue
        to_exit = bf - max(bn, _f(0.0))
        if sampled > to_exit:
            t = t + (to_exit + eps)
            if t_enter + t > t_exit:
                break
            bmax, blo, bhi = block_info(o + d * (t_enter + t + eps))
            continue
        t = t + sampled
        if t_enter + t > t_exit:
            break
        dens = _f(sample_grid(v, rng, (o + d * (t_enter + t)).astype(np.float64))) * _f(v["density"])
        if transmittance:
            with np.errstate(all="ignore"):
                T = T * (_f(1.0) - dens / maxd)
            pr = T
            if _f(rng.uf()) > pr:
                return 0.0
            T = T / pr
        else:
            with np.errstate(all="ignore"):
                if dens / maxd < _f(rng.uf()):
                    continue                            # null collision
            return float(t_enter + t)
    return float(T) if transmittance else -1.0


def volumes_transmittance(S, o, d, rng=None):   # CalculateVolumesTransmittance, Volume.slang:419-446
    T = 1.0
    for v in S.volumes:
        near, far = box_hit(o, d, v["lo"], v["hi"])
        near = max(near, 0.0)
        if v["grid"] is not None and far >= 0.0:      # heterogeneous: tracked (draws random numbers)
            T *= hetero_walk(S, v, o, d, rng, near, far, True)
            if T <= 0.0:
                return 0.0
        elif far - near > 0.0:
            T *= np.exp(-v["density"] * (far - near))
    return min(max(T, 0.0), 1.0)


def scattered_in_volume(S, pay, rng, P):   # RayGen.slang:162-270 without an atmosphere
    o, d = pay["origin"], pay["direction"]
    n = len(S.volumes)
    dist = [max(0.0, box_hit(o, d, v["lo"], v["hi"])[0]) for v in S.volumes]
    idx = list(range(n))
    for i in range(n):
        for j in range(i + 1, n):
            if dist[j] < dist[i]:
                dist[i], dist[j] = dist[j], dist[i]; idx[i], idx[j] = idx[j], idx[i]
    # GetDistanceToGeometry (RTCommon.slang:86-101, ray queries): the direction as it is; the TraceRay form (:103-117): normalised, TMax 1000
    h = S.closest(o, d, 0.00001, 1000000.0) if (P.flags & FLAG_RAY_QUERIES) else S.closest(o, _norm(d), 0.00001, 1000.0)
    dtg = h[0] if h is not None else -1.0
    sd, sv = -1.0, -1
    for i in range(n):
        v = S.volumes[idx[i]]
        near, far = box_hit(o, d, v["lo"], v["hi"])
        t = -1.0
        if not (far < 0.0) and not (sd >= 0.0 and near > sd):
            inside = far - max(near, 0.0)
            if inside > 0.0 and v["grid"] is not None:
                t = hetero_walk(S, v, o, d, rng, near, far, False)
            elif inside > 0.0:
                with np.errstate(all="ignore"):
                    sampled = -np.log(rng.uf()) / v["density"]
                if sampled < inside:
                    t = max(near, 0.0) + sampled
        if t >= 0.0 and (t < sd or sd < 0.0):
            sd, sv = t, idx[i]
    cchan, comp = pay["cchan"], -1
    if S.atm is not None:
        if cchan == -1:
            pick = rng.uf()
            cchan = 0 if pick < 0.33333 else (1 if pick < 0.66666 else 2)
        ta, comp = atm_scatter_distance(S.atm, rng, o, d, cchan)
        if ta >= 0.0 and (ta < sd or sd < 0.0):
            sd, sv = ta, -2
    if not (sd >= 0.0 and (dtg < 0.0 or sd < dtg)):
        return False
    if sv == -2:
        pay["cchan"] = cchan
        atmosphere_event(S, pay, rng, P, sd, comp)
        return True
    # ---- EvaluateVolumeScatteringEvent (RayGen.slang:272-380)
    v = S.volumes[sv]
    pay["origin"] = o + d * sd
    pay["emitted"] = v["emissive"].copy()
    po = pay["origin"]
    to_sky, sky = sample_sky(S, rng, P)
    sky[:3] = sky[:3] * P.sky_intensity
    if shadow_hit(S, P, po, to_sky)[0] is not None:
        sky = np.zeros(4)
    light = np.zeros(4); to_light = np.zeros(3)
    if S.emissive:
        to_light, light, e_inst, ti = sample_emissive(S, po, rng)
        h2, ids_defined = shadow_hit(S, P, po, to_light)
        hit_ids = S.ids[h2[3]] if h2 is not None else (0, 0)       # hitInstanceIndex / hitTriangleIdx default to 0 (RTCommon.slang:49-50)
        if hit_ids != (e_inst, ti) or not ids_defined:
            light = np.zeros(4)
    vd = pay["vdepth"]
    nd = sample_phase(S, v, d, vd, rng)
    ph = eval_phase(S, v, d, nd)
    if sky[3] > 0.0:
        pk = eval_phase(S, v, d, to_sky)
        T = volumes_transmittance(S, po, to_sky, rng)
        if S.atm is not None:
            T = T * nee_atmosphere_transmittance(S, rng, po, to_sky, pay["cchan"])
        if pk > 0.0:
            pay["emitted"] = pay["emitted"] + T * (v["color"] * pk) * (sky[:3] / sky[3]) * power_heuristics(sky[3], pk)
    if light[3] > 0.0:
        pl = eval_phase(S, v, d, to_light)
        T = volumes_transmittance(S, po, to_light, rng)
        if pl > 0.0:
            pay["emitted"] = pay["emitted"] + T * (v["color"] * pl) * (light[:3] / light[3]) * power_heuristics(light[3], pl)
    pay["direction"] = nd; pay["bxdf"] = v["color"] * ph; pay["pdf"] = ph
    pay["depth"] += 1; pay["vdepth"] = vd + 1
    return True


def sample_emissive(S, position, rng):
    """SampleEmissiveTriangle (Sampler.slang:348-422) -> (toLight, rgb | pdf, instance, triangle)."""
    ne = len(S.emissive)
    mi = min(int(np.floor(rng.uf() * ne)), ne - 1)
    e_inst = S.emissive[mi]
    e_mesh, e_mat, eM, _ = S.inst[e_inst]
    ev, eidx = S.meshes[e_mesh]
    ntri = len(eidx) // 3
    ti = min(int(np.floor(rng.uf() * ntri)), ntri - 1)
    tri = eidx.reshape(-1, 3)[ti]
    a0, a1, a2 = (eM[:3, :3] @ ev["position"][int(k)].astype(np.float64) + eM[:3, 3] for k in tri)
    x0, x1 = rng.uf(), rng.uf()
    su1 = np.sqrt(x0); b0 = 1.0 - su1; b1 = x1 * su1; b2 = 1.0 - b0 - b1
    tp = b0 * a0 + b1 * a1 + b2 * a2
    to_light = _norm(tp - position)
    ln = _norm(np.cross(a2 - a0, a1 - a0))
    area = float(np.linalg.norm(np.cross(a1 - a0, a2 - a0))) * 0.5
    d2 = float(np.dot(tp - position, tp - position)); ct = abs(float(np.dot(ln, to_light)))
    with np.errstate(all="ignore"):
        pdf = d2 / (float(ne) * float(ntri) * area * ct)
    t0, t1, t2 = (ev["texcoord"][int(k)].astype(np.float64) for k in tri)
    rgb = np.array(S.materials[e_mat]["emissive_color"], np.float64) * S.tex(S.materials[e_mat]["emissive_texture"], b0 * t0 + b1 * t1 + b2 * t2)[:3]
    return to_light, np.append(rgb, pdf), e_inst, ti


# ---------------------------------------------------------------- atmosphere
C_RAYLEIGH = np.array([5.802, 13.558, 33.100]) * 1e-6
C_MIE_S, C_MIE_A = 3.996e-6, 4.40e-6
C_OZONE = np.array([0.650, 1.881, 0.085]) * 1e-6
f32 = np.float32


def intersect_sphere(o, d, centre, radius):   # RTCommon.slang:168-185, float32 term for term
    ro = (o.astype(f32) - centre.astype(f32)).astype(f32)
    dd = d.astype(f32)
    dot = lambda a, b: f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))
    a = dot(dd, dd); b = f32(f32(2.0) * dot(ro, dd)); c = f32(dot(ro, ro) - f32(f32(radius) * f32(radius)))
    disc = f32(f32(b * b) - f32(f32(f32(4.0) * a) * c))
    if disc < 0:
        return -1.0, -1.0
    sq = f32(np.sqrt(disc))
    return float(f32(f32(-b - sq) / f32(f32(2.0) * a))), float(f32(f32(-b + sq) / f32(f32(2.0) * a)))


def atm_height(A, p):   # length(position - planet) - radius, float32
    q = (p.astype(f32) - np.array(A.planet_position[:], f32)).astype(f32)
    l = f32(np.sqrt(f32(f32(f32(q[0] * q[0]) + f32(q[1] * q[1])) + f32(q[2] * q[2]))))
    return float(f32(l - f32(A.planet_radius)))


def atm_densities(A, h, ch):
    r = np.exp(-h / A.rayleigh_density_falloff) * C_RAYLEIGH[ch] * A.rayleigh_multiplier[ch]
    m = np.exp(-h / A.mie_density_falloff) * (C_MIE_S + C_MIE_A) * A.mie_multiplier[ch]
    o = np.exp(-(abs(h - A.ozone_peak) / A.ozone_density_falloff)) * C_OZONE[ch] * A.ozone_multiplier[ch]
    return r, m, o


def atm_majorant(A, ch):
    r, m, _ = atm_densities(A, 0.0, ch)
    _, _, o = atm_densities(A, A.ozone_peak, ch)
    return r + m + o


def atm_transmittance(A, rng, o, d, ch):   # CalculateTransmittanceThroughAtmosphere: ratio tracking with roulette -> float3
    centre = np.array(A.planet_position[:], np.float64)
    if intersect_sphere(o, d, centre, A.planet_radius)[1] > 0.0:
        return np.zeros(3)
    t0, t1 = intersect_sphere(o, d, centre, A.planet_radius + A.atmosphere_height)
    tmin, tmax = max(t0, 0.0), t1
    if tmax < 0.0:
        return np.ones(3)
    maj = atm_majorant(A, ch)
    if maj <= 0.0:
        return np.ones(3)
    t, T = 0.0, 1.0
    for _ in range(1000):
        t += -np.log(1.0 - rng.uf()) / maj
        if t >= tmax - tmin:
            break
        h = atm_height(A, o + d * (t + tmin))
        if h < 0.0:
            break
        r, m, oz = atm_densities(A, h, ch)
        T *= 1.0 - (r + m + oz) / maj
        if rng.uf() > T:
            T = 0.0
            break
        T /= T
    out = np.zeros(3); out[ch] = T
    return out


def atm_scatter_distance(A, rng, o, d, ch):   # SampleAtmosphereScatterDistance: delta tracking -> (t | -1, component)
    centre = np.array(A.planet_position[:], np.float64)
    a0, a1 = intersect_sphere(o, d, centre, A.planet_radius + A.atmosphere_height)
    t_min_a, t_max_a = max(a0, 0.0), a1
    p0, _ = intersect_sphere(o, d, centre, A.planet_radius)
    if t_max_a < 0.0:
        return -1.0, -1
    maj = atm_majorant(A, ch)
    if maj <= 0.0:
        return -1.0, -1
    t = t_min_a
    for _ in range(1000):
        t += -np.log(1.0 - rng.uf()) / maj
        if t >= t_max_a:
            break
        if p0 > 0.0 and t >= p0:
            break
        r, m, oz = atm_densities(A, atm_height(A, o + d * t), ch)
        dens = r + m + oz
        if dens / maj < rng.uf():
            continue
        x = rng.uf()
        return t, (0 if x <= r / dens else (1 if x <= r / dens + m / dens else 2))
    return -1.0, -1


def sample_sun(S, rng, P):   # SampleSunDisk(0.004675) (Sampler.slang:430-463) -> (direction, rgb | pdf)
    A = S.atm
    sun = rotate(np.array([0.0, 0.0, -1.0]), np.array([1.0, 0.0, 0.0]), P.sky_altitude / 180.0 * np.pi)
    sun = rotate(sun, np.array([0.0, 1.0, 0.0]), P.sky_azimuth / 180.0 * np.pi)
    ctm = np.cos(0.004675)
    phi = 2.0 * np.pi * rng.uf()
    ct = ctm + (1.0 - ctm) * rng.uf()
    st = np.sqrt(1.0 - ct * ct)
    w = _norm(sun)
    up = np.array([0.0, 0.0, 1.0]) if abs(w[2]) < 0.999 else np.array([1.0, 0.0, 0.0])
    u = _norm(np.cross(up, w)); v = np.cross(w, u)
    d = u * (np.cos(phi) * st) + v * (np.sin(phi) * st) + w * ct
    return d, np.append(2e5 * np.array(A.sun_color[:], np.float64) * P.sky_intensity, 1.0 / (2.0 * np.pi * (1.0 - ctm)))


def sample_rayleigh(d, rng):   # Sampler.slang:194-214
    r0, r1 = rng.uf(), rng.uf()
    u = -np.cbrt(2.0 * (2.0 * r0 - 1.0) + np.sqrt(4.0 * (2.0 * r0 - 1.0) ** 2 + 1.0))
    ct = u - 1.0 / u
    phi = 2.0 * np.pi * r1
    st = np.sqrt(1.0 - ct * ct)
    nd = np.array([st * np.cos(phi), st * np.sin(phi), ct])
    up = np.array([0.0, 1.0, 0.0]) if abs(d[1]) < 0.9999999 else np.array([0.0, 0.0, 1.0])
    t = _norm(np.cross(up, d)); b = np.cross(d, t)
    return _norm(nd[0] * t + nd[1] * b + nd[2] * d)


def sample_sky(S, rng, P):   # ImportanceSampleSky (Sampler.slang:465-476)
    return sample_sun(S, rng, P) if S.atm is not None else sample_env(S, rng, P)


def nee_atmosphere_transmittance(S, rng, o, d, cchan):   # ClosestHit.slang:335-349 == RayGen.slang:328-343
    A = S.atm
    if cchan == -1:
        return np.array([atm_transmittance(A, rng, o, d, 0)[0], atm_transmittance(A, rng, o, d, 1)[1], atm_transmittance(A, rng, o, d, 2)[2]])
    return atm_transmittance(A, rng, o, d, cchan)


def atmosphere_event(S, pay, rng, P, sd, comp):   # EvaluateAtmosphereScatteringEvent with ENABLE_SKY_MIS (RayGen.slang:382-443)
    d = pay["direction"]
    pay["origin"] = pay["origin"] + sd * d
    nd = sample_rayleigh(d, rng) if comp == 0 else (sample_hg(d, 0.85, rng) if comp == 1 else d)
    to_sun, cp = sample_sun(S, rng, P)
    cp[:3] = cp[:3] * P.sky_intensity
    if shadow_hit(S, P, pay["origin"], to_sun)[0] is None:
        T = atm_transmittance(S.atm, rng, pay["origin"], to_sun, pay["cchan"]) * volumes_transmittance(S, pay["origin"], to_sun, rng)
    else:
        T = np.zeros(3)
    ray_phase = lambda a, b: (3.0 / (16.0 * np.pi)) * (1.0 + float(np.dot(a, b)) ** 2)
    if comp == 0:
        pay["emitted"] = pay["emitted"] + ray_phase(d, to_sun) * T * (cp[:3] / cp[3])
        pay["bxdf"] = np.full(3, ray_phase(d, nd)); pay["pdf"] = ray_phase(d, nd)
    elif comp == 1:
        pay["emitted"] = pay["emitted"] + phase_hg(d, to_sun, 0.85) * T * (cp[:3] / cp[3])
        pay["bxdf"] = np.full(3, phase_hg(d, nd, 0.85) * (1.0 - C_MIE_A / (C_MIE_S + C_MIE_A))); pay["pdf"] = phase_hg(d, nd, 0.85)
    else:
        pay["bxdf"] = np.zeros(3); pay["pdf"] = 1.0
    pay["direction"] = nd
    pay["depth"] += 1


def rotate(v, axis, theta):   # RTCommon.slang:37-45
    a = _norm(axis)
    return v * np.cos(theta) + np.cross(a, v) * np.sin(theta) + a * np.dot(a, v) * (1.0 - np.cos(theta))


def sample_env(S, rng, P):
    """ImportanceSampleEnvMap (Sampler.slang:286-346): three draws -> (direction, rgb * intensity | pdf)."""
    x0, x1, x2 = rng.uf(), rng.uf(), rng.uf()
    w, h = S.env_w, S.env_h
    size = w * h
    idx = min(int(np.float32(x0) * np.float32(size)), size - 1)   # uint(xi.x * float(size)): a float32 product
    if x1 < S.importance[idx]:
        ei = idx; x1 = x1 / S.importance[idx]
    else:
        ei = int(S.alias[idx]); x1 = (x1 - S.importance[idx]) / (1.0 - S.importance[idx])
    px_, py_ = ei % w, ei // w
    u = (px_ + x1) / w
    phi = u * (2.0 * np.pi) - np.pi
    step = np.pi / h
    th0 = py_ * step
    ct = np.cos(th0) * (1.0 - x2) + np.cos(th0 + step) * x2
    th = np.arccos(np.clip(ct, -1.0, 1.0))
    st = np.sin(th)
    v = th / np.pi
    d = np.array([np.sin(phi) * st, -ct, -np.cos(phi) * st])
    d = rotate(d, np.array([0.0, 1.0, 0.0]), P.sky_azimuth / 180.0 * np.pi)
    d = rotate(d, np.array([1.0, 0.0, 0.0]), P.sky_altitude / 180.0 * np.pi)
    val = S.env_lookup(u, v)
    return d, np.append(val[:3] * P.sky_intensity, val[3])


def miss(S, pay, P):   # Miss.slang with SHOW_ENV_MAP_DIRECTLY
    if S.atm is not None:   # :11-14: the sky is in-scattered sunlight only
        pay["depth"] = MAX_DEPTH
        return
    d = rotate(pay["direction"], np.array([1.0, 0.0, 0.0]), -(P.sky_altitude / 180.0 * np.pi))
    d = rotate(d, np.array([0.0, 1.0, 0.0]), -(P.sky_azimuth / 180.0 * np.pi))
    gamma = np.arcsin(np.clip(d[1], -1.0, 1.0)); theta = np.arctan2(d[0], -d[2])      # DirectionToUV (RTCommon.slang:129-136)
    if (P.flags & FLAG_SHOW_ENV_DIRECTLY) or pay["depth"] > 0:
        color_pdf = S.env_lookup(theta / np.pi * 0.5 + 0.5, gamma / np.pi + 0.5)
    else:
        color_pdf = np.array([0.0, 0.0, 0.0, 1.0])
    pay["emitted"] = color_pdf[:3] * P.sky_intensity
    if P.flags & FLAG_FURNACE:
        pay["emitted"] = np.ones(3)
    if (P.flags & FLAG_SKY_MIS) and pay["depth"] > 0:
        pay["emitted"] = pay["emitted"] * power_heuristics(pay["pdf"], color_pdf[3])
    pay["depth"] = MAX_DEPTH


def closest_hit(S, luts, pay, rd, hit, rng, P):
    t, hu, hv, gid = hit
    inst_id, prim = S.ids[gid]
    mesh, mat_id, M, Minv = S.inst[inst_id]
    vert, idx = S.meshes[mesh]
    i1, i2, i3 = (int(k) for k in idx.reshape(-1, 3)[prim])
    P1, P2, P3 = (vert["position"][k].astype(np.float64) for k in (i1, i2, i3))
    N1, N2, N3 = (vert["normal"][k].astype(np.float64) for k in (i1, i2, i3))
    b = np.array([1.0 - hu - hv, hu, hv])
    uv = sum(vert["texcoord"][k].astype(np.float64) * w_ for k, w_ in zip((i1, i2, i3), b))
    # ---- Surface.Initialize
    pos = M[:3, :3] @ (P1 * b[0] + P2 * b[1] + P3 * b[2]) + M[:3, 3]
    Ng = _norm(np.cross(P2 - P1, P3 - P1)); Ng = _norm(Ng @ Minv)      # mul(n, WorldToObject): row vector times the inverse
    geo = bool(P.flags & FLAG_GEOMETRY_NORMALS)
    N = Ng.copy() if geo else _norm(_norm(N1 * b[0] + N2 * b[1] + N3 * b[2]) @ Minv)
    view = -rd
    inside = bool(np.dot(Ng, view) < 0.0)
    if inside:
        N, Ng = -N, -Ng
    up = np.array([0.0, 0.0, 1.0]) if abs(N[2]) < 0.9999999 else np.array([1.0, 0.0, 0.0])
    T = _norm(np.cross(up, N)); B = _norm(np.cross(N, T))
    md = S.materials[mat_id]
    if not geo:
        nm = S.tex(md["normal_texture"], uv)[:3] * 2.0 - 1.0              # (the default normal map's texel is (128, 128, 255) / 255)
        N = _norm(nm[0] * T + nm[1] * B + nm[2] * N)                      # TangentToWorld
    if np.dot(N, view) < 0.0:
        N = _norm(N - view * (np.dot(N, view) - 0.01))
    refl = _norm(-view - 2.0 * np.dot(N, -view) * N)
    if np.dot(refl, Ng) < 0.0:
        N = _norm(N + Ng * (0.1 + np.dot(N, Ng)))
    T = _norm(np.cross(N, up)); B = _norm(np.cross(N, T))
    # ---- Material.Initialize (Material.slang:39-87) + RotateTangents
    md = dict(md)
    md["base_color"] = np.array(md["base_color"], np.float64) * S.tex(md["base_color_texture"], uv)[:3] ** 2.2
    md["roughness"] = md["roughness"] * S.tex(md["roughness_texture"], uv)[0]
    md["metallic"] = md["metallic"] * S.tex(md["metallic_texture"], uv)[0]
    emissive = np.array(md["emissive_color"], np.float64) * S.tex(md["emissive_texture"], uv)[:3]
    if P.flags & FLAG_FURNACE:   # Material.slang:78-86
        md["base_color"] = np.ones(3); emissive = np.zeros(3); md["specular_color"] = (1.0, 1.0, 1.0); md["medium_color"] = (1.0, 1.0, 1.0)
    m = Mat64(md, luts, inside=inside, ec=bool(P.flags & FLAG_ENERGY_COMPENSATION))
    is_light = bool((emissive > 0).any())
    rot = md["anisotropy_rotation"] * (np.pi / 180.0)
    T = T * np.cos(rot) + np.cross(N, T) * np.sin(rot) + N * np.dot(N, T) * (1.0 - np.cos(rot))
    B = np.cross(T, N)
    w2t = lambda v: _norm(np.array([np.dot(v, T), np.dot(v, B), np.dot(v, N)]))
    t2w = lambda v: _norm(v[0] * T + v[1] * B + v[2] * N)
    # ---- medium walk (ClosestHit.slang:80-116)
    if pay["in_medium"]:
        dist = float(np.linalg.norm(pay["origin"] - pos))
        if pay["med_aniso"] == 1.0:
            pay["bxdf"] = np.exp(-(1.0 - np.array(md["medium_color"], np.float64)) * md["medium_density"] * dist)
        else:
            with np.errstate(all="ignore"):
                sd = -np.log(rng.uf()) / pay["med_density"]
            if sd < dist:   # still inside the medium: a scattering event, then back to the sample loop (depth, pdf and emission untouched)
                pay["origin"] = pay["origin"] + sd * pay["direction"]
                pay["direction"] = sample_hg(pay["direction"], pay["med_aniso"], rng)
                pay["bxdf"] = np.array(pay["med_color"], np.float64)
                return
    # ---- sky NEE (ClosestHit.slang:118-147): sample, then the visibility test from pos + N * 1e-5
    can_sky = False; sky = np.zeros(4); to_sky = np.zeros(3); to_sky_t = np.zeros(3)
    if P.flags & FLAG_SKY_MIS:
        to_sky, sky = sample_sky(S, rng, P)
        sky[:3] = sky[:3] * P.sky_intensity            # the intensity is applied a second time here (:131), as upstream does
        to_sky_t = w2t(to_sky) if np.isfinite(to_sky).all() and np.abs(to_sky).max() > 0 else np.zeros(3)
        can_sky = shadow_hit(S, P, pos + N * 1e-5, to_sky)[0] is None
        if not can_sky:
            sky = np.zeros(4)
    # ---- light NEE (Sampler.slang:348-422)
    can_light = False; light_rgb = np.zeros(3); light_pdf = 0.0; to_light_t = None
    if (P.flags & FLAG_MESH_MIS) and not is_light and S.emissive:
        to_light, lc, e_inst, ti = sample_emissive(S, pos, rng)
        light_rgb, light_pdf = lc[:3], float(lc[3])
        if light_pdf > 0.0:
            to_light_t = w2t(to_light)
            h2, ids_defined = shadow_hit(S, P, pos + to_light * 1e-2, to_light)   # RTCommon.slang:54-60 (ray queries) / :64-84
            can_light = h2 is not None and ids_defined and S.ids[h2[3]] == (e_inst, ti)
            if not can_light:
                light_rgb = np.zeros(3); light_pdf = 0.0
    # ---- BSDF sampling
    V = w2t(_norm(-rd))
    L = sample64(m, V, rng)
    if L is None:
        sL, s_f, s_pdf = np.zeros(3), np.zeros(3), 0.0
    else:
        f, p = m.evaluate(V, L[None, :]); sL, s_f, s_pdf = L, f[0], float(p[0])
    refracted = bool(sL[2] < 0.0)
    with np.errstate(all="ignore"):
        scatter_w = t2w(sL)
    if not refracted and np.dot(scatter_w, Ng) < 0.0:
        s_pdf = 0.0; s_f = np.zeros(3)
    if refracted and inside:
        pay["in_medium"] = False
    elif refracted and not inside:
        pay["in_medium"] = True
        pay["med_color"] = np.array(md["medium_color"], np.float64); pay["med_aniso"] = md["medium_anisotropy"]; pay["med_density"] = md["medium_density"]
    k_f, k_pdf = np.zeros(3), 0.0
    if can_sky:
        f, p = m.evaluate(V, to_sky_t[None, :]); k_f, k_pdf = f[0], float(p[0])
    l_f, l_pdf = np.zeros(3), 0.0
    if can_light and not is_light:
        f, p = m.evaluate(V, to_light_t[None, :]); l_f, l_pdf = f[0], float(p[0])
    # ---- payload
    if not (P.flags & FLAG_MESH_MIS):
        pay["emitted"] = pay["emitted"] + emissive
    elif pay["depth"] == 0 and is_light:
        pay["emitted"] = pay["emitted"] + emissive
    elif is_light:
        w1, w2_, w3 = (M[:3, :3] @ q + M[:3, 3] for q in (P1, P2, P3))
        area = float(np.linalg.norm(np.cross(w2_ - w1, w3 - w1))) * 0.5
        d2 = float(np.dot(pos - pay["origin"], pos - pay["origin"]))
        ct = abs(float(np.dot(N, _norm(pay["origin"] - pos))))
        ntri = len(idx) // 3
        with np.errstate(all="ignore"):
            lp = (1.0 / len(S.emissive)) * (1.0 / ntri) * (1.0 / area) * (d2 / ct)
        lp = max(lp, P.emissive_pdf_bias)
        pay["emitted"] = pay["emitted"] + emissive * power_heuristics(pay["pdf"], lp)
    pay["origin"] = pos + N * (-1e-3 if refracted else 1e-3)
    pay["direction"] = scatter_w
    pay["bxdf"] = s_f; pay["pdf"] = s_pdf
    if can_sky:   # the transmittance is taken from the NEW origin (ClosestHit.slang:332-349); its draws happen whether or not it is used
        T_sky = volumes_transmittance(S, pay["origin"], to_sky, rng)
        if S.atm is not None:
            T_sky = T_sky * nee_atmosphere_transmittance(S, rng, pay["origin"], to_sky, pay["cchan"])
        if sky[3] > 0.0 and k_pdf > 0.0:
            pay["emitted"] = pay["emitted"] + (k_f * T_sky * sky[:3] / sky[3]) * power_heuristics(sky[3], k_pdf)
    if not is_light and can_light and light_pdf > 0.0 and l_pdf > 0.0:
        pay["emitted"] = pay["emitted"] + (l_f * volumes_transmittance(S, pay["origin"], to_light, rng) * light_rgb / light_pdf) * power_heuristics(light_pdf, l_pdf)
    invalid = s_pdf <= 0.0
    pay["depth"] = MAX_DEPTH + pay["depth"] if invalid else pay["depth"] + 1
