"""Regenerates the committed fixtures from the reference's assets (run in the build container only;
/root/reference does not exist on the GPU box).

  tests/golden/cornell_box.npz, cornell_box_glass.npz, viking_room.npz
      <- /root/reference/Assets/{CornellBox,CornellBoxGlass,VikingRoom}.gltf through scenes.load_gltf
  vulkan-path-tracer_amd/assets/lookup_tables.bin
      <- /root/reference/Assets/LookupTables/*.bin (fp32, index x + y*SX + z*SX*SY; LookupReflect.slang:32)
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference/Assets"


def save_gltf(s, path):
    """Writes a Scene as glTF 2.0 (+ .bin, + PNGs) such that scenes.load_gltf / the C++ SceneLoader give it back:
    undoes the Y flip and the winding swap of the loaders."""
    import json
    from PIL import Image
    base = os.path.splitext(path)[0]
    blob = bytearray()
    views, accessors = [], []

    def add(arr, ctype, typ):
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": arr.nbytes})
        blob.extend(arr.tobytes())
        acc = {"bufferView": len(views) - 1, "componentType": ctype, "count": len(arr), "type": typ}
        if typ == "VEC3" and ctype == 5126:
            acc["min"], acc["max"] = [float(v) for v in arr.min(0)], [float(v) for v in arr.max(0)]
        accessors.append(acc)
        return len(accessors) - 1

    flip = np.array([1, -1, 1], np.float32)
    meshes = []
    images, textures = [], []
    tex_of = {}

    def tex_ref(idx):
        if idx not in tex_of:
            t = s.textures[idx]
            name = "%s_tex%d.png" % (os.path.basename(base), idx)
            img = t if t.shape[2] == 4 else np.concatenate([t, t, t, np.full_like(t, 255)], 2)
            Image.fromarray(img, "RGBA").save(os.path.join(os.path.dirname(path), name))
            images.append({"uri": name})
            textures.append({"source": len(images) - 1})
            tex_of[idx] = len(textures) - 1
        return {"index": tex_of[idx]}

    mats = []
    for m in s.materials:
        pbr = {"baseColorFactor": [float(np.float32(c)) for c in m["base_color"]] + [1.0], "metallicFactor": float(np.float32(m["metallic"])),
               "roughnessFactor": float(np.float32(m["roughness"]))}
        g = {"name": m.get("name", ""), "pbrMetallicRoughness": pbr, "emissiveFactor": [float(np.float32(c)) for c in m["emissive_color"]],
             "extensions": {"KHR_materials_ior": {"ior": float(np.float32(m["ior"]))},
                            "KHR_materials_transmission": {"transmissionFactor": float(np.float32(m["transmission"]))},
                            "KHR_materials_specular": {"specularColorFactor": [float(np.float32(c)) for c in m["specular_color"]]}}}
        if m["base_color_texture"] != 0:
            pbr["baseColorTexture"] = tex_ref(m["base_color_texture"])
        if m["roughness_texture"] != 2:
            assert m["metallic_texture"] == m["roughness_texture"], "glTF has one metallicRoughness texture"
            pbr["metallicRoughnessTexture"] = tex_ref(m["roughness_texture"])
        if m["normal_texture"] != 1:
            g["normalTexture"] = tex_ref(m["normal_texture"])
        if m["emissive_texture"] != 4:
            g["emissiveTexture"] = tex_ref(m["emissive_texture"])
        mats.append(g)
    nodes = []
    for me, ma, x in s.instances:
        v, idx = s.meshes[me]
        prim = {"attributes": {"POSITION": add(np.ascontiguousarray(v["position"] * flip), 5126, "VEC3"),
                               "NORMAL": add(np.ascontiguousarray(v["normal"] * flip), 5126, "VEC3"),
                               "TEXCOORD_0": add(np.ascontiguousarray(v["texcoord"]), 5126, "VEC2")},
                "indices": add(np.ascontiguousarray(idx.reshape(-1, 3)[:, [0, 2, 1]].reshape(-1)), 5125, "SCALAR"), "material": ma}
        meshes.append({"primitives": [prim]})
        F = np.diag([1.0, -1.0, 1.0, 1.0])
        M = F @ np.asarray(x, np.float64) @ F
        nodes.append({"mesh": len(meshes) - 1, "matrix": [float(c) for c in M.T.reshape(-1)]})
    F = np.diag([1.0, -1.0, 1.0, 1.0])
    C = F @ np.asarray(s.view_inverse, np.float64) @ F
    nodes.append({"camera": 0, "matrix": [float(c) for c in C.T.reshape(-1)]})
    g = {"asset": {"version": "2.0", "generator": "tests/golden/make_fixtures.py"}, "scene": 0, "scenes": [{"nodes": list(range(len(nodes)))}], "nodes": nodes,
         "cameras": [{"type": "perspective", "perspective": {"aspectRatio": float(s.aspect), "yfov": 0.7853981633974483, "znear": 0.1, "zfar": 100.0}}],
         "materials": mats, "meshes": meshes, "accessors": accessors, "bufferViews": views,
         "buffers": [{"uri": os.path.basename(base) + ".bin", "byteLength": len(blob)}]}
    if images:
        g["images"], g["textures"] = images, textures
    open(base + ".bin", "wb").write(bytes(blob))
    json.dump(g, open(path, "w"), indent=1)


def textured_scene(pkg):
    """Small procedural scene with a PNG base-colour texture and a metallic-roughness texture: exercises the loaders'
    texture path (PNG decode, single-channel extraction) without shipping the 1 MB VikingRoom texture twice."""
    rng = np.random.RandomState(21)
    s = pkg.Scene()
    s.name = "textured_boxes"
    y, x = np.mgrid[0:64, 0:64]
    base = np.zeros((64, 64, 4), np.uint8)
    base[..., 0] = 40 + 3 * x; base[..., 1] = 30 + 3 * y; base[..., 2] = (((x // 8) + (y // 8)) % 2) * 180 + 40; base[..., 3] = 255
    mr = rng.randint(40, 255, (32, 32, 1)).astype(np.uint8)
    tb, tm = s.add_texture(base), s.add_texture(mr)
    s.materials.append(pkg.material(name="painted", base_color=(0.9, 0.9, 0.9), roughness=0.8, metallic=0.6, base_color_texture=tb, roughness_texture=tm, metallic_texture=tm))
    s.materials.append(pkg.material(name="lamp", emissive_color=(12.0, 10.0, 8.0)))
    s.materials.append(pkg.material(name="floor", base_color=(0.5, 0.55, 0.6), roughness=0.9))

    def quad(p0, du, dv, n):
        pos = np.array([p0, p0 + du, p0 + du + dv, p0 + dv], np.float32)
        return pos, np.tile(np.array(n, np.float32), (4, 1)), np.array([[0, 0], [2, 0], [2, 2], [0, 2]], np.float32), np.array([[0, 1, 2], [0, 2, 3]], np.uint32)
    faces = [((-1, -1, 1), (2, 0, 0), (0, 2, 0), (0, 0, 1)), ((1, -1, -1), (-2, 0, 0), (0, 2, 0), (0, 0, -1)), ((1, -1, 1), (0, 0, -2), (0, 2, 0), (1, 0, 0)),
             ((-1, -1, -1), (0, 0, 2), (0, 2, 0), (-1, 0, 0)), ((-1, -1, -1), (2, 0, 0), (0, 0, 2), (0, -1, 0)), ((-1, 1, 1), (2, 0, 0), (0, 0, -2), (0, 1, 0))]
    P, N, U, I = [], [], [], []
    for k, (p0, du, dv, n) in enumerate(faces):
        p, nn, u, i = quad(np.array(p0, np.float32), np.array(du, np.float32), np.array(dv, np.float32), n)
        P.append(p); N.append(nn); U.append(u); I.append(i + 4 * k)
    pos, nrm, uv, idx = np.concatenate(P), np.concatenate(N), np.concatenate(U), np.concatenate(I)
    g = np.cross(pos[idx][:, 1] - pos[idx][:, 0], pos[idx][:, 2] - pos[idx][:, 0])
    bad = (g * nrm[idx][:, 0]).sum(1) < 0
    idx[bad] = idx[bad][:, [0, 2, 1]]
    cube = s.add_mesh(pos, nrm, uv, idx)
    s.add_instance(cube, 0, pkg._xform((-1.4, -1.0, 0.0), (0.9, 1.0, 0.9), ry=0.5))
    s.add_instance(cube, 0, pkg._xform((1.3, -0.6, -0.8), (0.6, 0.6, 0.6), ry=-0.3))
    s.add_instance(cube, 1, pkg._xform((0.0, -3.4, 0.5), (0.5, 0.05, 0.5)))
    s.add_instance(cube, 2, pkg._xform((0.0, 0.1, 0.0), (6.0, 0.1, 6.0)))
    s.env = np.zeros((1, 1, 4), np.float32)
    s.view_inverse = (np.linalg.inv(pkg.look_at((0.5, -2.2, 6.5), (0.0, -0.9, 0.0), (0, -1, 0))) @ np.diag([1.0, -1.0, 1.0, 1.0])).astype(np.float32)
    return s


def main():
    pkg = importlib.import_module("vulkan-path-tracer_amd.scenes")
    out = os.path.dirname(os.path.abspath(__file__))
    for src, dst in (("CornellBox", "cornell_box"), ("CornellBoxGlass", "cornell_box_glass"), ("VikingRoom", "viking_room")):
        s = pkg.load_gltf(os.path.join(REF, src + ".gltf"))
        s.save(os.path.join(out, dst + ".npz"))
        print(dst, "tris", s.triangle_count(), "materials", len(s.materials), "textures", len(s.textures))
    # glTF test assets for the C++ SceneLoader (vulkan-path-tracer_amd/host) and scenes.load_gltf
    save_gltf(pkg.load_gltf(os.path.join(REF, "CornellBox.gltf")), os.path.join(out, "cornell_box.gltf"))
    save_gltf(textured_scene(pkg), os.path.join(out, "textured_boxes.gltf"))
    lt = os.path.join(REF, "LookupTables")
    np.concatenate([np.fromfile(os.path.join(lt, n), "<f4") for n in
                    ("ReflectionLookup.bin", "RefractionLookupHitFromOutside.bin", "RefractionLookupHitFromInside.bin")]
                   ).tofile(os.path.join(ROOT, "vulkan-path-tracer_amd", "assets", "lookup_tables.bin"))


if __name__ == "__main__":
    main()
