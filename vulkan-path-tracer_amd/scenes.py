"""Scene containers and builders for tests / bench (host-side plumbing, numpy only).

A `Scene` is exactly what `vpt_scene_desc` (include/vpt.h) carries: meshes (32-byte vertices + u32
indices), 112-byte materials, instances {mesh, material, mat4}, RGBA8/R8 textures, an RGBA32F
environment image and the three energy-compensation tables — the data PathTracer::SetScene assembles
from VulkanHelper::AssetImporter output (reference PathTracer.cpp:158-676).

`load_gltf` is a minimal glTF 2.0 reader standing in for the absent VulkanHelper/assimp importer
(parity unpinned): node TRS hierarchy, cameras, pbrMetallicRoughness,
KHR_materials_{emissive_strength,transmission,ior,specular}.  The reference's world is Y-down
(Sampler.slang:336, RTCommon.slang:129-136, FlyCamera.cpp:52-56), glTF is Y-up, so every node matrix
M becomes F·M·F with F = diag(1,-1,1,1), vertices/normals are mirrored and the winding is swapped so
geometric and vertex normals stay on the same side.
"""
import ctypes as C
import json
import os

import numpy as np

from . import _abi

VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("texcoord", "<f4", 2)])
assert VERTEX_DTYPE.itemsize == 32
ASSET_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")


def material(**kw):
    """PathTracer::Material defaults (PathTracer.h:14-33)."""
    m = dict(base_color=(1, 1, 1), emissive_color=(0, 0, 0), specular_color=(1, 1, 1), medium_color=(1, 1, 1),
             medium_emissive_color=(0, 0, 0), metallic=0.0, roughness=1.0, ior=1.5, transmission=0.0, anisotropy=0.0,
             anisotropy_rotation=0.0, medium_density=0.0, medium_anisotropy=0.0, base_color_texture=0,
             normal_texture=1, roughness_texture=2, metallic_texture=3, emissive_texture=4, name="")
    for k, v in kw.items():
        if k not in m:
            raise KeyError(k)
        m[k] = v
    return m


def default_textures():
    """LoadDefaultTexture (PathTracer.cpp:1557-1582) in first-use order of SetScene's texture loop
    (PathTracer.cpp:227-332): base(white RGBA), normal (128,128,255,255), roughness R8 255, metallic R8 255,
    emissive (white RGBA)."""
    w = np.full((1, 1, 4), 255, np.uint8)
    n = np.array([[[128, 128, 255, 255]]], np.uint8)
    r = np.full((1, 1, 1), 255, np.uint8)
    return [w.copy(), n, r.copy(), r.copy(), w.copy()]


_LUT_CACHE = {}


def load_luts():
    """The three reference tables (Assets/LookupTables/*.bin), shipped as a data asset."""
    if not _LUT_CACHE:
        z = np.load(os.path.join(ASSET_DIR, "lookup_tables.npz"))
        _LUT_CACHE["r"] = np.ascontiguousarray(z["reflection"], np.float32)
        _LUT_CACHE["o"] = np.ascontiguousarray(z["refraction_outside"], np.float32)
        _LUT_CACHE["i"] = np.ascontiguousarray(z["refraction_inside"], np.float32)
    return _LUT_CACHE["r"], _LUT_CACHE["o"], _LUT_CACHE["i"]


class Scene:
    def __init__(self):
        self.meshes = []      # list of (vertices[VERTEX_DTYPE], indices[u32])
        self.materials = []   # list of dict (material())
        self.instances = []   # list of (mesh_index, material_index, 4x4 float32 math matrix)
        self.textures = default_textures()  # list of uint8 [h,w,c]
        self.env = np.zeros((1, 1, 4), np.float32)  # RGBA32F [h,w,4]
        self.luts = None
        self.view_inverse = np.eye(4, dtype=np.float32)
        self.aspect = 16.0 / 9.0
        self.name = "scene"

    # ---- helpers
    def add_mesh(self, positions, normals, uvs, indices):
        v = np.zeros(len(positions), VERTEX_DTYPE)
        v["position"] = positions
        v["normal"] = normals
        if uvs is not None:
            v["texcoord"] = uvs
        self.meshes.append((v, np.ascontiguousarray(indices, np.uint32).reshape(-1)))
        return len(self.meshes) - 1

    def add_texture(self, arr):
        arr = np.ascontiguousarray(arr, np.uint8)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        self.textures.append(arr)
        return len(self.textures) - 1

    def add_instance(self, mesh, mat, xform=None):
        m = np.eye(4, dtype=np.float32) if xform is None else np.asarray(xform, np.float32)
        self.instances.append((int(mesh), int(mat), m))

    def triangle_count(self):
        return sum(len(self.meshes[m][1]) // 3 for m, _, _ in self.instances)

    def projection_inverse(self, aspect=None):
        """inverse(glm::perspective(radians(45), aspect, 0.1, 100)) — PathTracer.cpp:578."""
        return np.linalg.inv(perspective(45.0, self.aspect if aspect is None else aspect, 0.1, 100.0)).astype(np.float32)

    def default_size(self):
        """PathTracer.cpp:509-511: 1080 rows, width = (uint)(1080 * aspect)."""
        return int(np.float32(1080.0) * np.float32(self.aspect)), 1080

    # ---- C-ABI view
    def to_desc(self):
        """Returns (vpt_scene_desc, keepalive)."""
        keep = []
        meshes = (_abi.Mesh * len(self.meshes))()
        for i, (v, idx) in enumerate(self.meshes):
            v = np.ascontiguousarray(v)
            idx = np.ascontiguousarray(idx, np.uint32)
            keep += [v, idx]
            meshes[i] = _abi.Mesh(v.ctypes.data, len(v), idx.ctypes.data, len(idx))
        mats = (_abi.Material * len(self.materials))()
        for i, m in enumerate(self.materials):
            mm = mats[i]
            for k in ("base_color", "emissive_color", "specular_color", "medium_color", "medium_emissive_color"):
                getattr(mm, k)[:] = [float(x) for x in m[k]]
            for k in ("metallic", "roughness", "ior", "transmission", "anisotropy", "anisotropy_rotation",
                      "medium_density", "medium_anisotropy"):
                setattr(mm, k, float(m[k]))
            for k in ("base_color_texture", "normal_texture", "roughness_texture", "metallic_texture",
                      "emissive_texture"):
                setattr(mm, k, int(m[k]))
        insts = (_abi.Instance * len(self.instances))()
        for i, (me, ma, x) in enumerate(self.instances):
            insts[i].mesh_index = me
            insts[i].material_index = ma
            insts[i].transform[:] = [float(v) for v in np.asarray(x, np.float32).T.reshape(-1)]  # column-major
        texs = (_abi.Texture * len(self.textures))()
        for i, t in enumerate(self.textures):
            t = np.ascontiguousarray(t, np.uint8)
            keep.append(t)
            texs[i] = _abi.Texture(t.shape[1], t.shape[0], t.shape[2], t.ctypes.data)
        env = np.ascontiguousarray(self.env, np.float32)
        lr, lo, li = self.luts if self.luts is not None else load_luts()
        keep += [meshes, mats, insts, texs, env, lr, lo, li]
        d = _abi.SceneDesc()
        d.meshes = meshes
        d.mesh_count = len(self.meshes)
        d.materials = mats
        d.material_count = len(self.materials)
        d.instances = insts
        d.instance_count = len(self.instances)
        d.textures = texs
        d.texture_count = len(self.textures)
        d.env_rgba = env.ctypes.data
        d.env_width = env.shape[1]
        d.env_height = env.shape[0]
        d.lut_reflection = lr.ctypes.data
        d.lut_refraction_outside = lo.ctypes.data
        d.lut_refraction_inside = li.ctypes.data
        return d, keep

    # ---- fixtures (npz), so GPU-box tests never need /root/reference
    def save(self, path):
        d = {"n_mesh": len(self.meshes), "n_tex": len(self.textures), "env": self.env,
             "view_inverse": self.view_inverse, "aspect": np.float64(self.aspect), "name": self.name,
             "materials": json.dumps(self.materials),
             "inst_idx": np.array([(a, b) for a, b, _ in self.instances], np.uint32).reshape(-1, 2),
             "inst_xf": np.array([x for _, _, x in self.instances], np.float32).reshape(-1, 4, 4)}
        for i, (v, idx) in enumerate(self.meshes):
            d["mv%d" % i] = v
            d["mi%d" % i] = idx
        for i, t in enumerate(self.textures):
            d["tx%d" % i] = t
        np.savez_compressed(path, **d)

    @staticmethod
    def load(path):
        z = np.load(path, allow_pickle=False)
        s = Scene()
        s.meshes = [(z["mv%d" % i], z["mi%d" % i]) for i in range(int(z["n_mesh"]))]
        s.textures = [z["tx%d" % i] for i in range(int(z["n_tex"]))]
        s.env = z["env"]
        s.view_inverse = z["view_inverse"]
        s.aspect = float(z["aspect"])
        s.name = str(z["name"])
        s.materials = json.loads(str(z["materials"]))
        s.instances = [(int(a), int(b), x) for (a, b), x in zip(z["inst_idx"], z["inst_xf"])]
        return s


# ------------------------------------------------------------------ glm restatements (host side)
def perspective(fov_deg, aspect, near, far):
    """glm::perspective (RH, [-1,1] depth) as FlyCamera.cpp:92-94 / PathTracer.cpp:578 call it."""
    f = 1.0 / np.tan(np.radians(fov_deg) / 2.0)
    m = np.zeros((4, 4), np.float64)
    m[0, 0] = f / aspect
    m[1, 1] = f
    m[2, 2] = (far + near) / (near - far)
    m[2, 3] = 2.0 * far * near / (near - far)
    m[3, 2] = -1.0
    return m


def look_at(eye, center, up):
    """glm::lookAt (RH)."""
    eye, center, up = (np.asarray(a, np.float64) for a in (eye, center, up))
    f = center - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[0, 3], m[1, 3], m[2, 3] = -s @ eye, -u @ eye, f @ eye
    return m


def colmajor(m):
    """4x4 math matrix -> float[16] column-major (glm memory order) ctypes array."""
    return (C.c_float * 16)(*[float(v) for v in np.asarray(m, np.float32).T.reshape(-1)])


# ------------------------------------------------------------------ glTF reader
_FLIP = np.diag([1.0, -1.0, 1.0, 1.0])


def _node_matrix(n):
    if "matrix" in n:
        return np.array(n["matrix"], np.float64).reshape(4, 4).T
    t = np.eye(4)
    if "translation" in n:
        t[:3, 3] = n["translation"]
    r = np.eye(4)
    if "rotation" in n:
        x, y, z, w = n["rotation"]
        r[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    s = np.eye(4)
    if "scale" in n:
        s[0, 0], s[1, 1], s[2, 2] = n["scale"]
    return t @ r @ s


def load_gltf(path, image_loader=None):
    base = os.path.dirname(path)
    g = json.load(open(path))
    bufs = [open(os.path.join(base, b["uri"]), "rb").read() for b in g["buffers"]]

    def accessor(i):
        a = g["accessors"][i]
        bv = g["bufferViews"][a["bufferView"]]
        dt = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}[a["componentType"]]
        nc = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}[a["type"]]
        off = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = bv.get("byteStride", 0)
        isz = np.dtype(dt).itemsize * nc
        raw = bufs[bv["buffer"]]
        if stride and stride != isz:
            out = np.zeros((a["count"], nc), dt)
            for k in range(a["count"]):
                out[k] = np.frombuffer(raw, dt, nc, off + k * stride)
            return out
        return np.frombuffer(raw, dt, a["count"] * nc, off).reshape(a["count"], nc).copy()

    s = Scene()
    s.name = os.path.splitext(os.path.basename(path))[0]
    tex_cache = {}

    def texture_index(tex_ref, single_channel):
        if tex_ref is None:
            return None
        img = g["images"][g["textures"][tex_ref["index"]]["source"]]
        key = (img["uri"], single_channel)
        if key not in tex_cache:
            if image_loader is None:
                from PIL import Image
                arr = np.array(Image.open(os.path.join(base, img["uri"])).convert("RGBA"), np.uint8)
            else:
                arr = image_loader(os.path.join(base, img["uri"]))
            if single_channel:  # LoadTexture(..., onlySingleChannel=true) keeps R (PathTracer.cpp:826-836)
                arr = arr[:, :, :1]
            tex_cache[key] = s.add_texture(arr)
        return tex_cache[key]

    for m in g.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        ext = m.get("extensions", {})
        bc = pbr.get("baseColorFactor", [1, 1, 1, 1])
        em = np.array(m.get("emissiveFactor", [0, 0, 0]), np.float64)
        em = em * ext.get("KHR_materials_emissive_strength", {}).get("emissiveStrength", 1.0)
        mm = material(name=m.get("name", ""), base_color=tuple(bc[:3]), emissive_color=tuple(em),
                      metallic=pbr.get("metallicFactor", 1.0), roughness=pbr.get("roughnessFactor", 1.0),
                      ior=ext.get("KHR_materials_ior", {}).get("ior", 1.5),
                      transmission=ext.get("KHR_materials_transmission", {}).get("transmissionFactor", 0.0),
                      specular_color=tuple(ext.get("KHR_materials_specular", {}).get("specularColorFactor", [1, 1, 1])))
        for key, ref, single in (("base_color_texture", pbr.get("baseColorTexture"), False),
                                 ("normal_texture", m.get("normalTexture"), False),
                                 ("roughness_texture", pbr.get("metallicRoughnessTexture"), True),
                                 ("metallic_texture", pbr.get("metallicRoughnessTexture"), True),
                                 ("emissive_texture", m.get("emissiveTexture"), False)):
            ti = texture_index(ref, single)
            if ti is not None:
                mm[key] = ti
        s.materials.append(mm)
    if not s.materials:
        s.materials.append(material(name="default"))

    prim_mesh = {}
    for mi, m in enumerate(g.get("meshes", [])):
        for pi, p in enumerate(m["primitives"]):
            pos = accessor(p["attributes"]["POSITION"]).astype(np.float32)
            nrm = accessor(p["attributes"]["NORMAL"]).astype(np.float32) if "NORMAL" in p["attributes"] else np.zeros_like(pos)
            uv = accessor(p["attributes"]["TEXCOORD_0"]).astype(np.float32) if "TEXCOORD_0" in p["attributes"] else None
            idx = accessor(p["indices"]).astype(np.uint32).reshape(-1, 3) if "indices" in p else np.arange(len(pos), dtype=np.uint32).reshape(-1, 3)
            pos = pos * np.array([1, -1, 1], np.float32)
            nrm = nrm * np.array([1, -1, 1], np.float32)
            idx = idx[:, [0, 2, 1]]
            prim_mesh[(mi, pi)] = (s.add_mesh(pos, nrm, uv, idx), p.get("material", 0))

    cam = [None]

    def walk(ni, parent):
        n = g["nodes"][ni]
        M = parent @ _node_matrix(n)
        if "mesh" in n:
            for pi in range(len(g["meshes"][n["mesh"]]["primitives"])):
                me, ma = prim_mesh[(n["mesh"], pi)]
                s.add_instance(me, ma, (_FLIP @ M @ _FLIP).astype(np.float32))
        if "camera" in n and cam[0] is None:
            c = g["cameras"][n["camera"]]
            cam[0] = (_FLIP @ M @ _FLIP, c.get("perspective", {}).get("aspectRatio", 16.0 / 9.0))
        for ch in n.get("children", []):
            walk(ch, M)

    for ni in g["scenes"][g.get("scene", 0)]["nodes"]:
        walk(ni, np.eye(4))
    if cam[0] is None:  # PathTracer.cpp:171-178 default camera
        cam[0] = (np.linalg.inv(look_at((0, 0, 5), (0, 0, 0), (0, 1, 0))), 16.0 / 9.0)
    s.view_inverse = cam[0][0].astype(np.float32)
    s.aspect = float(np.float32(cam[0][1]))
    return s


# ------------------------------------------------------------------ environments
def constant_env(rgb=(1, 1, 1), w=64, h=32):
    e = np.zeros((h, w, 4), np.float32)
    e[:, :, :3] = rgb
    return e


def sun_sky_env(w, h, seed=7, sun_peak=5.0e4):
    """Analytic sun-and-sky HDR: gradient sky + ground + small sun disc."""
    rng = np.random.RandomState(seed)
    v = (np.arange(h, dtype=np.float64) + 0.5) / h
    u = (np.arange(w, dtype=np.float64) + 0.5) / w
    theta = v[:, None] * np.pi            # 0 = top of the image = world -Y = up
    phi = u[None, :] * 2 * np.pi - np.pi
    up = np.cos(theta)                    # +1 at zenith
    sky = np.stack([0.25 + 0.35 * (1 - np.clip(up, 0, 1)), 0.45 + 0.35 * (1 - np.clip(up, 0, 1)), 0.9 + 0 * up], -1)
    ground = np.stack([0.18 + 0 * up, 0.16 + 0 * up, 0.13 + 0 * up], -1)
    img = np.where((up > 0)[..., None], sky, ground) * np.ones((1, w, 1))
    sun_theta, sun_phi = np.radians(35.0 + 10 * rng.rand()), np.radians(-40.0 + 20 * rng.rand())
    d = np.stack([np.sin(phi) * np.sin(theta), -np.cos(theta) * np.ones_like(phi), -np.cos(phi) * np.sin(theta)], -1)
    sd = np.array([np.sin(sun_phi) * np.sin(sun_theta), -np.cos(sun_theta), -np.cos(sun_phi) * np.sin(sun_theta)])
    cosang = d @ sd
    sun = np.clip((cosang - np.cos(np.radians(1.5))) / (1 - np.cos(np.radians(1.5))), 0, 1)
    img = img + sun[..., None] * np.array([1.0, 0.95, 0.85]) * sun_peak
    e = np.zeros((h, w, 4), np.float32)
    e[:, :, :3] = img
    return e
