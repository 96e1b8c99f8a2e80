"""The C-ABI library loads and exports every symbol include/vpt.h declares — and NONE of the laboratory's (include/vpt_lab.h), which exist in
libvpt_hip_lab.so only (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(name="vpt.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vpt_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_known_entry_points():
    fns = header_functions()
    assert "vpt_create" in fns and "vpt_render" in fns and "vpt_postprocess" in fns and len(fns) >= 20


def test_library_exports_every_declared_symbol(vpt):
    lib = vpt.load_library()
    for name in header_functions():
        assert hasattr(lib, name), name
    lab = header_functions("vpt_lab.h")
    assert lab == sorted(vpt._abi.LAB_ENTRY_POINTS)
    # and the ctypes mirror covers both headers
    assert sorted(vpt._abi.PROTOTYPES) == sorted(header_functions() + lab)


def test_product_library_has_no_laboratory(vpt):
    """libvpt_hip.so: no vpt_lab_* symbol, no laboratory kernel, under 3 MB; VPT_PIPELINE_STAGED_R1 is refused at vpt_create (checked before any device is touched)."""
    import subprocess
    path = vpt.library_path(lab=False)
    vpt.build(lab=False)
    names = subprocess.check_output(["nm", "-D", path], text=True)
    assert "vpt_lab_" not in names and "vpt_create" in names
    blob = open(path, "rb").read()
    for kernel in (b"k_trace_pool", b"k_trace_pair", b"k_trace_base", b"k_extend", b"k_connect", b"k_raygenE"):
        assert kernel not in blob, kernel
    assert len(blob) < 3 * 1024 * 1024, len(blob)
    lib = C.CDLL(path)
    lib.vpt_create.restype = C.c_void_p
    err = C.c_int(0)
    cfg = vpt._abi.Config(0, 16, 16, 0, 1, 0, 0, 0, vpt._abi.PIPELINE_STAGED_R1, 0, 0)
    assert not lib.vpt_create(C.byref(cfg), C.byref(err)) and err.value == -6   # VPT_ERR_UNSUPPORTED


def test_struct_layouts_match_reference_contract(vpt):
    a = vpt._abi
    assert C.sizeof(a.Material) == 112          # PathTracer.h:12-34, SURVEY §7.2
    assert vpt.scenes.VERTEX_DTYPE.itemsize == 32  # Bindings.slang:7-12
    assert C.sizeof(a.Instance) == 72
    assert C.sizeof(a.Ray) == 32 and C.sizeof(a.Hit) == 20  # ray in 32 B, hit out 20 B
    assert C.sizeof(a.Params) == 13 * 4
    assert C.sizeof(a.PostParams) == 28          # the six fields of PostProcessor.h:8-21 + the schedule selector


def test_defaults_match_reference(vpt):
    lib = vpt.load_library()
    p = vpt._abi.Params()
    lib.vpt_default_params(C.byref(p))
    q = vpt.default_params()
    for k, _ in vpt._abi.Params._fields_:
        assert getattr(p, k) == getattr(q, k), k
    assert (p.samples_per_frame, p.max_samples, p.max_depth, p.max_luminance) == (1, 5000, 200, 500.0)  # PathTracer.h:202-205
    pp = vpt._abi.PostParams()
    lib.vpt_default_post_params(C.byref(pp))
    assert (pp.exposure, round(pp.gamma, 5), pp.bloom_threshold, pp.bloom_strength, pp.mip_count, pp.falloff_range) == (1.0, 2.2, 2.0, 1.0, 10, 5.0)


def test_bad_arguments_return_error_codes(vpt):
    lib = vpt.load_library()
    err = C.c_int(0)
    assert not lib.vpt_create(None, C.byref(err)) and err.value == -1
    cfg = vpt._abi.Config(0, 0, 0, 0, 1, 0, 0, 0, 0)
    assert not lib.vpt_create(C.byref(cfg), C.byref(err)) and err.value == -1
    assert lib.vpt_render(None, 1, None) == -1
    assert lib.vpt_get_stats(None, None) == -1


def test_no_device_means_failure_not_fallback(vpt):
    """Without a usable HIP device the backend refuses to exist; it never routes to a CPU path."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(vpt.VptError, match="NO_DEVICE"):
        vpt.PathTracer(16, 16)


def test_ctypes_mirror_matches_the_compiled_header(vpt, tmp_path):
    """sizeof and every field offset of each C struct, as gcc lays out include/vpt.h, against the ctypes mirror."""
    import subprocess
    a = vpt._abi
    pairs = [("vpt_material", a.Material), ("vpt_volume", a.Volume), ("vpt_atmosphere", a.Atmosphere), ("vpt_mesh", a.Mesh), ("vpt_instance", a.Instance),
             ("vpt_texture", a.Texture), ("vpt_scene_desc", a.SceneDesc), ("vpt_params", a.Params), ("vpt_post_params", a.PostParams),
             ("vpt_config", a.Config), ("vpt_stats", a.Stats), ("vpt_ray", a.Ray), ("vpt_hit", a.Hit), ("vpt_comm_info", a.CommInfo)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vpt.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf(" %%zu", offsetof(%s, %s));' % (cname, fname))
        lines.append('printf("\\n");')
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), str(src), "-o", exe])
    out = subprocess.check_output([exe], text=True).strip().splitlines()
    for (cname, cls), line in zip(pairs, out):
        tok = line.split()
        assert tok[0] == cname and int(tok[1]) == C.sizeof(cls), (cname, tok[1], C.sizeof(cls))
        for (fname, _), off in zip(cls._fields_, tok[2:]):
            assert getattr(cls, fname).offset == int(off), (cname, fname)
    # the flag bits the Python side names
    hdr = open(os.path.join(ROOT, "include", "vpt.h")).read()
    for name, val in (("VPT_FLAG_SKY_MIS", a.FLAG_SKY_MIS), ("VPT_FLAG_MESH_MIS", a.FLAG_MESH_MIS), ("VPT_FLAG_LOCAL_HITS", a.FLAG_LOCAL_HITS)):
        m = re.search(r"#define %s \(1u << (\d+)\)" % name, hdr)
        assert m and (1 << int(m.group(1))) == val, name
