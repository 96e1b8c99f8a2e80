"""The C++ host facade (vulkan-path-tracer_amd/host: PathTracer / PostProcessor / FlyCamera / SceneLoader over the
C-ABI).  CPU part: it builds, its glTF importer agrees byte-for-byte with scenes.load_gltf, its glm restatements hold.
GPU part: rendering through the facade's CLI equals the oracle on the same scene and camera."""
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "vulkan-path-tracer_amd", "host")
CLI = os.path.join(HOST, "vpt_render")
GOLDEN = os.path.join(ROOT, "tests", "golden")
LUTS = os.path.join(ROOT, "vulkan-path-tracer_amd", "assets", "lookup_tables.bin")


@pytest.fixture(scope="module")
def cli(vpt):
    vpt.load_library()  # libvpt_hip.so must exist to link against
    import fcntl
    with open(os.path.join(HOST, ".build.lock"), "w") as lock:   # pytest-xdist workers share the tree: one make at a time (a second one would relink the binary under a running test)
        fcntl.flock(lock, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-C", HOST, "vpt_render"], stdout=subprocess.DEVNULL)
    return CLI


def read_dump(path):
    b = open(path, "rb").read()
    p = [0]

    def u32():
        v = int(np.frombuffer(b, "<u4", 1, p[0])[0]); p[0] += 4; return v

    def take(n):
        v = b[p[0]:p[0] + n]; p[0] += n; return v
    meshes = []
    for _ in range(u32()):
        nv, ni = u32(), u32()
        meshes.append((take(nv * 32), take(ni * 4)))
    mats = take(u32() * 112)
    inst = [(u32(), u32(), np.frombuffer(take(64), "<f4").reshape(4, 4).T.copy()) for _ in range(u32())]
    texs = []
    for _ in range(u32()):
        w, h, c = u32(), u32(), u32()
        texs.append((w, h, c, take(w * h * c)))
    cams = [(float(np.frombuffer(take(4), "<f4")[0]), np.frombuffer(take(64), "<f4").reshape(4, 4).T.copy()) for _ in range(u32())]
    assert p[0] == len(b)
    return meshes, mats, inst, texs, cams


@pytest.mark.parametrize("name", ["cornell_box", "textured_boxes", "textured_boxes_jpg"])
def test_cpp_importer_equals_python_loader(cli, vpt, tmp_path, name):
    import ctypes as C
    gltf = os.path.join(GOLDEN, name + ".gltf")
    info = json.loads(subprocess.check_output([cli, "--scene", gltf, "--dump-scene", str(tmp_path / "s.bin")]))
    sc = vpt.scenes.load_gltf(gltf)
    assert info["triangles"] == sc.triangle_count() and info["materials"] == len(sc.materials) and info["textures"] == len(sc.textures)
    meshes, mats, inst, texs, cams = read_dump(str(tmp_path / "s.bin"))
    assert len(meshes) == len(sc.meshes)
    for (vb, ib), (v, i) in zip(meshes, sc.meshes):
        assert vb == np.ascontiguousarray(v).tobytes() and ib == np.ascontiguousarray(i, np.uint32).tobytes()
    desc, keep = sc.to_desc()
    assert mats == C.string_at(desc.materials, 112 * len(sc.materials))
    for (me, ma, x), (pme, pma, px) in zip(inst, sc.instances):
        assert (me, ma) == (pme, pma) and np.array_equal(x, np.asarray(px, np.float32))
    for (w, h, c, d), t in zip(texs, sc.textures):
        assert (h, w, c) == t.shape and d == np.ascontiguousarray(t).tobytes()
    assert abs(cams[0][0] - sc.aspect) < 1e-6 and np.allclose(cams[0][1], sc.view_inverse, atol=1e-5)


def test_cpp_camera_and_matrix_restatements(cli):
    r = json.loads(subprocess.check_output([cli, "--selftest"]))
    assert r["view_err"] < 1e-4 and r["proj_err"] < 1e-5 and r["inverse_err"] < 1e-5
    assert abs(r["fov"] - 45.0) < 1e-3 and abs(r["aspect"] - 16 / 9) < 1e-4
    assert r["up_dy"] != 0.0
    assert r["move_keeps_state"] is True      # volumes, phase function, atmosphere, shard identity all travel with a move


def test_cli_reports_errors_without_crashing(cli, tmp_path):
    p = subprocess.run([cli, "--scene", str(tmp_path / "missing.gltf"), "--info"], capture_output=True)
    assert p.returncode == 1 and b"cannot open" in p.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("gpus", [2, 3])
def test_multi_gpu_cli_is_bit_identical_to_one_device(cli, tmp_path, gpus):
    """vpt_render --gpus N: N shard contexts (one host thread each), vpt_multi_gather_shards into shard 0, post there.
    On a one-GPU box every shard is pinned to device 0 (--devices): the gather is then a same-device peer copy, the rest
    of the path (partition, threads, gather into root, re-interleave, post on root) is the one an 8-GPU node runs."""
    gltf = os.path.join(GOLDEN, "textured_boxes.gltf")
    outs = []
    for n in (1, gpus):
        rad, ppm = str(tmp_path / ("r%d.f32" % n)), str(tmp_path / ("o%d.ppm" % n))
        out = json.loads(subprocess.check_output([cli, "--scene", gltf, "--luts", LUTS, "--size", "161x91", "--spp", "6", "--depth", "5", "--radiance", rad, "--ppm", ppm,
                                                  "--gpus", str(n), "--devices", ",".join(["0"] * n)]))
        assert out["gpus"] == n and out["samples"] == 6
        outs.append((np.fromfile(rad, "<f4"), open(ppm, "rb").read()))
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]


@pytest.mark.gpu
@pytest.mark.parametrize("name,extra", [("cornell_box", []), ("textured_boxes", []), ("textured_boxes_jpg", []), ("cornell_box", ["--split", "2"]),
                                        ("cornell_box", ["--split", "1", "--async"]), ("textured_boxes", ["--split", "1", "--async"]),
                                        # 4 dispatches per call against a budget of 5: the second call runs one AND reports done (GetSamplesAccumulated must say 5)
                                        ("cornell_box", ["--split", "1", "--async", "--async-step", "4"])])
def test_render_through_the_cpp_facade_matches_the_oracle(cli, vpt, oracle, tmp_path, name, extra):
    """(--async: PathTraceAsync + PostProcessAsync per frame with a one-frame fence lag, the reference's Editor loop — Editor.cpp:116,129;
    the fused pipeline's frames go over the lanes and captured graphs, the streams pipeline's are finished by the next call.)"""
    gltf = os.path.join(GOLDEN, name + ".gltf")
    rad, cam, ppm = str(tmp_path / "r.f32"), str(tmp_path / "c.f32"), str(tmp_path / "o.ppm")
    w, h, spp, depth, seed = 160, 90, 5, 6, 3
    out = json.loads(subprocess.check_output([cli, "--scene", gltf, "--luts", LUTS, "--size", "%dx%d" % (w, h), "--spp", str(spp), "--depth", str(depth),
                                              "--seed", str(seed), "--radiance", rad, "--camera", cam, "--ppm", ppm] + extra))
    assert out["samples"] == spp and (out["width"], out["height"]) == (w, h)
    img = np.fromfile(rad, "<f4").reshape(h, w, 4)
    m = np.fromfile(cam, "<f4").reshape(2, 4, 4)
    sc = vpt.scenes.load_gltf(gltf)
    split = int(extra[1]) if extra else 1
    o = oracle.Oracle(sc, w, h)
    o.set_camera(m[0].T, m[1].T)  # column-major dumps -> math matrices
    o.set_params(vpt.default_params(max_depth=depth, base_seed=seed, screen_chunk_count=split, max_samples=spp))
    o.render(spp * split * split)
    ref = o.radiance(); o.close()
    assert np.array_equal(img, ref)
    ref8, _ = oracle.postprocess(ref, vpt.default_post_params())
    body = open(ppm, "rb").read().split(b"\n255\n", 1)[1]
    assert np.array_equal(np.frombuffer(body, np.uint8).reshape(h, w, 3), ref8[..., :3])


@pytest.mark.gpu
@pytest.mark.parametrize("which,kind,size", [("reflect", 0, (64, 64, 32)), ("refract-above", 1, (32, 32, 8)), ("refract-below", 2, (32, 32, 8))])
def test_lookup_table_calculator_facade(cli, vpt, tmp_path, which, kind, size):
    """LookupTableCalculator::New(shader, defines).CalculateTable(size, samples) through the CLI == vpt_lut_calculate."""
    path = str(tmp_path / "t.bin")
    out = json.loads(subprocess.check_output([cli, "--make-lut", which, "--lut-samples", "400", "--lut-time-seed", "9", "--lut-size", "%dx%dx%d" % size,
                                              "--lut-out", path]))
    assert out["size"] == list(size) and out["samples_per_cell"] == 400
    got = np.fromfile(path, "<f4")
    assert np.array_equal(got, vpt.calculate_lut(kind, size, 400, time_ms=9).reshape(-1))
    p = subprocess.run([cli, "--make-lut", which, "--lut-samples", "5", "--lut-out", path], capture_output=True)
    assert p.returncode == 1 and b"vpt_lut_calculate" in p.stderr


@pytest.mark.gpu
def test_volumes_through_the_cpp_facade(cli, vpt, oracle, tmp_path):
    """AddVolume / SetPhaseFunction on the facade == vpt_set_volumes / vpt_set_phase_function == the oracle."""
    gltf = os.path.join(GOLDEN, "cornell_box.gltf")
    rad, cam = str(tmp_path / "r.f32"), str(tmp_path / "c.f32")
    w, h, spp, depth = 128, 72, 3, 6
    subprocess.check_output([cli, "--scene", gltf, "--luts", LUTS, "--size", "%dx%d" % (w, h), "--spp", str(spp), "--depth", str(depth), "--radiance", rad,
                             "--camera", cam, "--volume", "-5,-10.5,-5,5,-0.5,5,0.15,0.4,0.9,0.8,0.7", "--volume", "-1,-4,-1,1,-2,1,2.0,-0.3,0.2,0.3,0.4",
                             "--phase", "draine"])
    img = np.fromfile(rad, "<f4").reshape(h, w, 4)
    m = np.fromfile(cam, "<f4").reshape(2, 4, 4)
    o = oracle.Oracle(vpt.scenes.load_gltf(gltf), w, h)
    o.set_camera(m[0].T, m[1].T)
    o.set_params(vpt.default_params(max_depth=depth, base_seed=1, max_samples=spp))
    o.set_volumes([vpt.volume(corner_min=(-5, -10.5, -5), corner_max=(5, -0.5, 5), density=0.15, anisotropy=0.4, color=(0.9, 0.8, 0.7)),
                   vpt.volume(corner_min=(-1, -4, -1), corner_max=(1, -2, 1), density=2.0, anisotropy=-0.3, color=(0.2, 0.3, 0.4))])
    o.set_phase_function(vpt.PHASE_DRAINE)
    o.render(spp)
    ref = o.radiance(); o.close()
    assert np.array_equal(img, ref)


def test_hdr_reader_and_png_writer_agree_between_cpp_and_python(cli, vpt, tmp_path):
    """the .hdr env-map reader (flat and RLE scanlines) and the PNG writer, C++ vs Python, byte for byte."""
    rng = np.random.default_rng(4)
    sky = vpt.scenes.sun_sky_env(96, 48, seed=9, sun_peak=3.0e4)[..., :3].copy()
    sky[5:9, 10:40] = 0.0                       # exponent-0 texels and long runs
    sky[20, :] = sky[20, 0]                      # a constant scanline: one run per component
    sky[30:34] *= rng.uniform(1e-6, 1e3, (4, 96, 1)).astype(np.float32)
    for rle in (True, False):
        path = str(tmp_path / ("sky_%d.hdr" % rle))
        vpt.imagefiles.save_hdr(path, sky, rle=rle)
        py = vpt.imagefiles.load_hdr(path)
        dump = str(tmp_path / "env.f32")
        out = json.loads(subprocess.check_output([cli, "--env-hdr", path, "--dump-env", dump]))
        assert (out["width"], out["height"]) == (96, 48)
        cpp = np.fromfile(dump, "<f4").reshape(48, 96, 4)
        assert np.array_equal(cpp, py)
        # RGBE keeps 8 mantissa bits of the brightest channel
        big = sky.max(axis=2) > 1e-30
        assert np.all(np.abs(py[..., :3][big] - sky[big]) <= sky.max(axis=2)[big][:, None] / 128.0 + 1e-30)
        assert np.all(py[..., 3] == 1.0)
    narrow = str(tmp_path / "narrow.hdr")       # width < 8: flat scanlines even when RLE is asked for
    vpt.imagefiles.save_hdr(narrow, sky[:3, :5], rle=True)
    assert vpt.imagefiles.load_hdr(narrow).shape == (3, 5, 4)
    p = subprocess.run([cli, "--env-hdr", os.path.join(GOLDEN, "cornell_box.gltf"), "--dump-env", str(tmp_path / "x")], capture_output=True)
    assert p.returncode == 1 and b"not a Radiance HDR" in p.stderr
    # PNG: C++ writer -> C++ reader (inside the CLI) and -> Python reader; Python writer -> same pixels
    png = str(tmp_path / "t.png")
    assert json.loads(subprocess.check_output([cli, "--png-roundtrip", png]))["png_roundtrip"] is True
    a = vpt.imagefiles.load_png(png)
    assert a.shape == (21, 37, 4)
    png2 = str(tmp_path / "t2.png")
    vpt.imagefiles.save_png(png2, a)
    assert np.array_equal(vpt.imagefiles.load_png(png2), a)
    for name in ("textured_boxes_base.png", "textured_boxes_rough.png"):
        f = os.path.join(GOLDEN, name)
        if os.path.exists(f):
            assert vpt.imagefiles.load_png(f).shape[2] == 4


@pytest.mark.gpu
def test_hdr_env_and_png_export_through_the_cli(cli, vpt, oracle, tmp_path):
    """SetEnvMapFilepath(.hdr) + Editor::SaveToFile(.png) end to end: radiance == oracle with the decoded env, PNG == oracle post."""
    gltf = os.path.join(GOLDEN, "cornell_box.gltf")
    hdr, rad, cam, png = (str(tmp_path / n) for n in ("sky.hdr", "r.f32", "c.f32", "o.png"))
    vpt.imagefiles.save_hdr(hdr, vpt.scenes.sun_sky_env(64, 32, seed=2, sun_peak=500.0)[..., :3])
    w, h, spp, depth = 96, 54, 3, 5
    subprocess.check_output([cli, "--scene", gltf, "--luts", LUTS, "--size", "%dx%d" % (w, h), "--spp", str(spp), "--depth", str(depth), "--radiance", rad,
                             "--camera", cam, "--env-hdr", hdr, "--png", png])
    img = np.fromfile(rad, "<f4").reshape(h, w, 4)
    m = np.fromfile(cam, "<f4").reshape(2, 4, 4)
    sc = vpt.scenes.load_gltf(gltf)
    sc.env = vpt.imagefiles.load_hdr(hdr)
    o = oracle.Oracle(sc, w, h)
    o.set_camera(m[0].T, m[1].T)
    o.set_params(vpt.default_params(max_depth=depth, base_seed=1, max_samples=spp))
    o.render(spp)
    ref = o.radiance(); o.close()
    assert np.array_equal(img, ref)
    ref8, _ = oracle.postprocess(ref, vpt.default_post_params())
    assert np.array_equal(vpt.imagefiles.load_png(png), ref8)


@pytest.mark.gpu
def test_atmosphere_through_the_cpp_facade(cli, vpt, oracle, tmp_path):
    """SetEnableAtmosphere + SetSkyAltitude/Azimuth on the facade == vpt_set_atmosphere == the oracle."""
    gltf = os.path.join(GOLDEN, "cornell_box.gltf")
    rad, cam = str(tmp_path / "r.f32"), str(tmp_path / "c.f32")
    w, h, spp, depth = 96, 54, 3, 6
    subprocess.check_output([cli, "--scene", gltf, "--luts", LUTS, "--size", "%dx%d" % (w, h), "--spp", str(spp), "--depth", str(depth), "--radiance", rad,
                             "--camera", cam, "--atmosphere", "--sun", "-35,120"])
    img = np.fromfile(rad, "<f4").reshape(h, w, 4)
    m = np.fromfile(cam, "<f4").reshape(2, 4, 4)
    o = oracle.Oracle(vpt.scenes.load_gltf(gltf), w, h)
    o.set_camera(m[0].T, m[1].T)
    o.set_params(vpt.default_params(max_depth=depth, base_seed=1, max_samples=spp, sky_altitude=-35.0, sky_azimuth=120.0))
    o.set_atmosphere(vpt.atmosphere())
    o.render(spp)
    ref = o.radiance(); o.close()
    assert np.array_equal(img, ref)


def _repack(src_gltf, dst, mode):
    """Rewrite a .gltf + .bin + PNG scene as (mode 'glb') one binary container with the images embedded as bufferViews, or
    (mode 'data') a .gltf whose buffer and images are base64 data: URIs."""
    import base64, struct
    base = os.path.dirname(src_gltf)
    g = json.load(open(src_gltf))
    blob = bytearray(open(os.path.join(base, g["buffers"][0]["uri"]), "rb").read())
    assert len(g["buffers"]) == 1
    if mode == "glb":
        for img in g.get("images", []):
            png = open(os.path.join(base, img.pop("uri")), "rb").read()
            while len(blob) % 4:
                blob.append(0)
            g["bufferViews"].append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(png)})
            img["bufferView"] = len(g["bufferViews"]) - 1; img["mimeType"] = "image/png"
            blob += png
        while len(blob) % 4:
            blob.append(0)
        g["buffers"][0] = {"byteLength": len(blob)}
        js = json.dumps(g).encode()
        js += b" " * (-len(js) % 4)
        body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(blob), 0x004E4942) + bytes(blob)
        open(dst, "wb").write(struct.pack("<4sII", b"glTF", 2, 12 + len(body)) + body)
    else:
        g["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(bytes(blob)).decode()
        for img in g.get("images", []):
            img["uri"] = "data:image/png;base64," + base64.b64encode(open(os.path.join(base, img["uri"]), "rb").read()).decode()
        json.dump(g, open(dst, "w"))


@pytest.mark.parametrize("mode", ["glb", "data"])
def test_glb_container_and_data_uris(cli, vpt, tmp_path, mode):
    """SURVEY 8f-3 loader breadth: the binary .glb container (JSON + BIN chunks, images as bufferViews) and base64 data: URIs
    import to exactly the scene of the plain .gltf + .bin + .png form, in the C++ importer and in the Python loader."""
    src = os.path.join(GOLDEN, "textured_boxes.gltf")
    dst = str(tmp_path / ("scene.glb" if mode == "glb" else "scene_inline.gltf"))
    _repack(src, dst, mode)
    a, b = vpt.scenes.load_gltf(src), vpt.scenes.load_gltf(dst)
    assert len(a.meshes) == len(b.meshes) and len(a.textures) == len(b.textures) and len(a.textures) > 5
    for (va, ia), (vb, ib) in zip(a.meshes, b.meshes):
        assert np.array_equal(va, vb) and np.array_equal(ia, ib)
    for ta, tb in zip(a.textures, b.textures):
        assert np.array_equal(ta, tb)
    assert [{k: v for k, v in m.items() if k != "name"} for m in a.materials] == [{k: v for k, v in m.items() if k != "name"} for m in b.materials]
    for name, path in (("a", src), ("b", dst)):
        subprocess.check_output([cli, "--scene", path, "--dump-scene", str(tmp_path / (name + ".bin"))])
    assert open(tmp_path / "a.bin", "rb").read() == open(tmp_path / "b.bin", "rb").read()
    # a truncated container is an error, not a crash
    if mode == "glb":
        bad = str(tmp_path / "bad.glb")
        open(bad, "wb").write(open(dst, "rb").read()[:200])
        p = subprocess.run([cli, "--scene", bad, "--info"], capture_output=True)
        assert p.returncode == 1 and b"glb" in p.stderr


def test_material_extensions_import_identically(cli, vpt, tmp_path):
    """Every material field the reference takes from its importer (PathTracer.cpp:338-348) has a glTF source: core PBR factors,
    KHR_materials_emissive_strength / ior / transmission / specular / anisotropy (rotation: radians -> the reference's degrees).
    C++ importer and Python loader produce the same 112-byte material records."""
    import ctypes as C
    doc, tri = _minimal()
    doc["materials"] = [
        {"name": "all", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.5, 0.25, 1.0], "metallicFactor": 0.25, "roughnessFactor": 0.375},
         "emissiveFactor": [1.0, 0.5, 0.25],
         "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 12.5}, "KHR_materials_ior": {"ior": 1.33},
                        "KHR_materials_transmission": {"transmissionFactor": 0.75}, "KHR_materials_specular": {"specularColorFactor": [0.9, 0.8, 0.7]},
                        "KHR_materials_anisotropy": {"anisotropyStrength": 0.6, "anisotropyRotation": 1.0}}},
        {"name": "negative rotation", "extensions": {"KHR_materials_anisotropy": {"anisotropyStrength": 0.3, "anisotropyRotation": -7.5}}},
        {"name": "plain"}]
    doc["meshes"][0]["primitives"][0]["material"] = 0
    doc["meshes"].append({"primitives": [{"attributes": {"POSITION": 0}, "indices": 1, "material": 1}]})
    doc["meshes"].append({"primitives": [{"attributes": {"POSITION": 0}, "indices": 1, "material": 2}]})
    doc["nodes"] = [{"mesh": 0}, {"mesh": 1}, {"mesh": 2}]; doc["scenes"] = [{"nodes": [0, 1, 2]}]
    path = _gltf(tmp_path, "ext.gltf", doc, tri)
    sc = vpt.scenes.load_gltf(path)
    m = sc.materials[0]
    assert abs(m["anisotropy"] - 0.6) < 1e-7 and abs(m["anisotropy_rotation"] - 57.29578) < 1e-4 and abs(m["ior"] - 1.33) < 1e-7
    assert abs(m["emissive_color"][0] - 12.5) < 1e-6 and abs(m["transmission"] - 0.75) < 1e-7 and tuple(m["specular_color"]) == (0.9, 0.8, 0.7)
    assert abs(sc.materials[1]["anisotropy_rotation"] - (360.0 - (7.5 * 180.0 / np.pi) % 360.0) % 360.0) < 1e-3 and 0 <= sc.materials[1]["anisotropy_rotation"] < 360
    subprocess.check_output([cli, "--scene", path, "--dump-scene", str(tmp_path / "s.bin")])
    _, mats, _, _, _ = read_dump(str(tmp_path / "s.bin"))
    desc, keep = sc.to_desc()
    assert mats == C.string_at(desc.materials, 112 * len(sc.materials))


def _gltf(tmp_path, name, doc, bin_bytes=b"\x00" * 64):
    import json as _json
    (tmp_path / "b.bin").write_bytes(bin_bytes)
    p = tmp_path / name
    p.write_text(doc if isinstance(doc, str) else _json.dumps(doc))
    return str(p)


def _minimal():
    import struct
    tri = struct.pack("<9f", 0, 0, 0, 1, 0, 0, 0, 1, 0) + struct.pack("<3H", 0, 1, 2) + b"\x00\x00"
    doc = {"asset": {"version": "2.0"}, "buffers": [{"uri": "b.bin", "byteLength": len(tri)}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": 36}, {"buffer": 0, "byteOffset": 36, "byteLength": 6}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"}, {"bufferView": 1, "componentType": 5123, "count": 3, "type": "SCALAR"}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1}]}], "nodes": [{"mesh": 0}], "scenes": [{"nodes": [0]}], "scene": 0}
    return doc, tri


def test_cpp_importer_rejects_malformed_files_without_crashing(cli, tmp_path):
    """the C++ glTF / PNG readers must treat file contents as untrusted (the Python twin raises in these cases)."""
    import copy
    doc, tri = _minimal()
    ok = subprocess.run([cli, "--scene", _gltf(tmp_path, "ok.gltf", doc, tri), "--info"], capture_output=True)
    assert ok.returncode == 0 and json.loads(ok.stdout)["triangles"] == 1
    bad = {}
    d = copy.deepcopy(doc); d["accessors"][0]["count"] = 2 ** 62; bad["huge_count"] = d
    d = copy.deepcopy(doc); d["accessors"][0]["count"] = 2 ** 40; d["bufferViews"][0]["byteStride"] = 2 ** 40; bad["stride_overflow"] = d
    d = copy.deepcopy(doc); d["bufferViews"][0]["byteStride"] = 1; bad["stride_below_element_size"] = d   # would let `count` outgrow the buffer it reads
    d = copy.deepcopy(doc); d["accessors"][0]["bufferView"] = 7; bad["bufferview_index"] = d
    d = copy.deepcopy(doc); d["bufferViews"][0]["buffer"] = 3; bad["buffer_index"] = d
    d = copy.deepcopy(doc); del d["accessors"]; bad["no_accessors"] = d
    d = copy.deepcopy(doc); del d["accessors"][0]["type"]; bad["no_type"] = d
    d = copy.deepcopy(doc); d["accessors"][0]["byteOffset"] = -5; bad["negative_offset"] = d
    d = copy.deepcopy(doc); d["nodes"] = [{"children": [1]}, {"children": [0], "mesh": 0}]; bad["node_cycle"] = d
    d = copy.deepcopy(doc); d["nodes"][0]["mesh"] = 9; bad["mesh_index"] = d
    d = copy.deepcopy(doc); d["scenes"][0]["nodes"] = [5]; bad["root_index"] = d
    d = copy.deepcopy(doc); d["materials"] = [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 4}}}]; bad["texture_index"] = d
    d = copy.deepcopy(doc); d["nodes"][0]["matrix"] = [1, 0, 0]; bad["short_matrix"] = d
    for name, dd in bad.items():
        p = subprocess.run([cli, "--scene", _gltf(tmp_path, name + ".gltf", dd, tri), "--info"], capture_output=True, timeout=30)
        assert p.returncode == 1 and p.stderr, (name, p.returncode, p.stderr[-200:])
    deep = "[" * 5000 + "]" * 5000
    p = subprocess.run([cli, "--scene", _gltf(tmp_path, "deep.gltf", '{"x": %s}' % deep, tri), "--info"], capture_output=True, timeout=30)
    assert p.returncode == 1
    # PNG readers: a short IHDR chunk and absurd dimensions
    import struct, zlib
    def chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    sig = b"\x89PNG\r\n\x1a\n"
    pngs = {"short_ihdr": sig + chunk(b"IHDR", b"\0" * 4) + chunk(b"IDAT", zlib.compress(b"\0" * 16)) + chunk(b"IEND", b""),
            "huge": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 2 ** 31 - 1, 2 ** 31 - 1, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 16)) + chunk(b"IEND", b"")}
    for name, data in pngs.items():
        (tmp_path / (name + ".png")).write_bytes(data)
        d = copy.deepcopy(doc); d["images"] = [{"uri": name + ".png"}]; d["textures"] = [{"source": 0}]
        d["materials"] = [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}}]
        p = subprocess.run([cli, "--scene", _gltf(tmp_path, name + ".gltf", d, tri), "--info"], capture_output=True, timeout=30)
        assert p.returncode == 1 and b"PNG" in p.stderr, (name, p.returncode, p.stderr[-200:])
