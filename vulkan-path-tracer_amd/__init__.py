"""MI355X wavefront render backend for the Vulkan-Path-Tracer hot path.

This package is a thin ctypes shim over the C-ABI in include/vpt.h (libvpt_hip.so, hand-written HIP for
gfx950).  All compute happens in the library; there is no Python or CPU fallback: if the library or a
HIP device is missing, loading / vpt_create fail loudly.
"""
import ctypes as C
import os

import numpy as np

from . import _abi, imagefiles, scenes  # noqa: F401
from ._abi import default_params, default_post_params, volume, atmosphere  # noqa: F401
from ._abi import PHASE_HENYEY_GREENSTEIN, PHASE_DRAINE, PHASE_HENYEY_GREENSTEIN_PLUS_DRAINE  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class VptError(RuntimeError):
    pass


def _want_lab():
    """VPT_LAB=1 in the environment selects the LABORATORY build (libvpt_hip_lab.so: include/vpt_lab.h, the measured-and-rejected kernel
    variants, round 1's stage kernels).  It chooses which LIBRARY is loaded — a measuring tool's choice — never a behaviour of the product library."""
    return os.environ.get("VPT_LAB", "0") not in ("", "0")


def library_path(lab=None):
    lab = _want_lab() if lab is None else lab
    return os.path.join(_HERE, "libvpt_hip_lab.so" if lab else "libvpt_hip.so")


def build(force=False, verbose=False, lab=None):
    from . import _build as _b
    return _b.build(force=force, verbose=verbose, lab=_want_lab() if lab is None else lab)


def load_library():
    """Loads libvpt_hip.so — or libvpt_hip_lab.so with VPT_LAB=1 — building it first when hipcc is available and sources are newer."""
    global _LIB
    if _LIB is None:
        path = library_path()
        try:
            path = build()
        except Exception as e:  # no hipcc on this box: the prebuilt in-tree .so must exist
            if not os.path.exists(path):
                raise VptError("%s is missing and cannot be built: %s" % (os.path.basename(path), e))
        _LIB = _abi.bind(C.CDLL(path))
    return _LIB


def has_lab():
    """True when the loaded library is the laboratory build (exports vpt_lab_*, accepts VPT_PIPELINE_STAGED_R1)."""
    return bool(load_library().has_lab)


def _check(lib, ctx, rc, what):
    if rc != 0:
        msg = lib.vpt_last_error(ctx).decode() if ctx else ""
        raise VptError("%s failed: %s %s" % (what, _abi.ERR_NAMES.get(rc, rc), msg))


LUT_REFLECT, LUT_REFRACT_ABOVE, LUT_REFRACT_BELOW = 0, 1, 2


def calculate_lut(kind, size, sample_count, time_ms=0, device=0):
    """LookupTableCalculator::CalculateTable (LookupTableCalculator.cpp:44) on the GPU: returns float32 [z, y, x].
    kind: LUT_REFLECT (LookupReflect.slang), LUT_REFRACT_ABOVE / _BELOW (LookupRefract.slang + define)."""
    lib = load_library()
    sx, sy, sz = (int(v) for v in size)
    out = np.zeros((sz, sy, sx), np.float32)
    _check(lib, None, lib.vpt_lut_calculate(device, kind, sx, sy, sz, sample_count, time_ms, out.ctypes.data), "vpt_lut_calculate")
    return out


class PathTracer:
    """Mirror of the reference's PathTracer + PostProcessor call surface over the C-ABI."""

    def __init__(self, width, height, device=0, shard_rank=0, shard_count=1, frames_in_flight=0, profile=False,
                 count_traversal=False, pipeline=0, build_flags=0, resident_frames=0):
        self.lib = load_library()
        cfg = _abi.Config(device, width, height, shard_rank, shard_count, frames_in_flight, int(profile), int(count_traversal), int(pipeline), int(build_flags), int(resident_frames))
        err = C.c_int(0)
        self.ctx = self.lib.vpt_create(C.byref(cfg), C.byref(err))
        if not self.ctx:
            raise VptError("vpt_create failed: %s (no CPU fallback exists)" % _abi.ERR_NAMES.get(err.value, err.value))
        self.width, self.height = width, height
        self.shard_rank, self.shard_count = shard_rank, shard_count

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.vpt_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- PathTracer API
    def set_scene(self, scene, set_camera=True):
        desc, keep = scene.to_desc()
        _check(self.lib, self.ctx, self.lib.vpt_set_scene(self.ctx, C.byref(desc)), "vpt_set_scene")
        del keep
        if set_camera:
            self.set_camera(scene.view_inverse, scene.projection_inverse(self.width / self.height))

    def set_camera(self, view_inverse, projection_inverse):
        _check(self.lib, self.ctx, self.lib.vpt_set_camera(self.ctx, scenes.colmajor(view_inverse), scenes.colmajor(projection_inverse)), "vpt_set_camera")

    def set_params(self, params):
        _check(self.lib, self.ctx, self.lib.vpt_set_params(self.ctx, C.byref(params)), "vpt_set_params")

    def set_volumes(self, volumes):
        """AddVolume / RemoveVolume / SetVolume (PathTracer.h:157-159): replaces the whole list."""
        arr = (_abi.Volume * max(len(volumes), 1))(*volumes)
        _check(self.lib, self.ctx, self.lib.vpt_set_volumes(self.ctx, arr, len(volumes)), "vpt_set_volumes")

    def add_density_grid(self, grid):
        """AddDensityDataToVolume with the .vdb already decoded: float32 [z, y, x] raw densities -> grid index."""
        g = np.ascontiguousarray(grid, np.float32)
        rc = self.lib.vpt_add_density_grid(self.ctx, g.shape[2], g.shape[1], g.shape[0], g.ctypes.data)
        if rc < 0:
            _check(self.lib, self.ctx, rc, "vpt_add_density_grid")
        return rc

    def clear_density_grids(self):
        _check(self.lib, self.ctx, self.lib.vpt_clear_density_grids(self.ctx), "vpt_clear_density_grids")

    def set_atmosphere(self, atm):
        """SetEnableAtmosphere(True) + the planet/density setters; None disables (PathTracer.h:168-179)."""
        _check(self.lib, self.ctx, self.lib.vpt_set_atmosphere(self.ctx, C.byref(atm) if atm is not None else None), "vpt_set_atmosphere")

    def set_phase_function(self, phase):
        _check(self.lib, self.ctx, self.lib.vpt_set_phase_function(self.ctx, phase), "vpt_set_phase_function")

    def set_material(self, index, mat):
        _check(self.lib, self.ctx, self.lib.vpt_set_material(self.ctx, index, C.byref(mat)), "vpt_set_material")

    def get_material(self, index):
        m = _abi.Material()
        _check(self.lib, self.ctx, self.lib.vpt_get_material(self.ctx, index, C.byref(m)), "vpt_get_material")
        return m

    def resize(self, w, h):
        _check(self.lib, self.ctx, self.lib.vpt_resize(self.ctx, w, h), "vpt_resize")
        self.width, self.height = w, h

    def reset(self):
        _check(self.lib, self.ctx, self.lib.vpt_reset(self.ctx), "vpt_reset")

    def render(self, dispatches):
        done = C.c_int(0)
        _check(self.lib, self.ctx, self.lib.vpt_render(self.ctx, dispatches, C.byref(done)), "vpt_render")
        return bool(done.value)

    # ---- the asynchronous per-frame form (PathTrace(cmd) / PostProcess(cmd) record and return; include/vpt.h)
    def render_async(self, dispatches):
        """vpt_render_async: enqueues and returns (done, ticket)."""
        done, ticket = C.c_int(0), C.c_uint64(0)
        _check(self.lib, self.ctx, self.lib.vpt_render_async(self.ctx, dispatches, C.byref(done), C.byref(ticket)), "vpt_render_async")
        return bool(done.value), int(ticket.value)

    def postprocess_device(self, post_params=None, rgba8_device=None):
        """vpt_postprocess_device: the post chain behind the render, RGBA8 left on the device (output_device()); returns the ticket."""
        pp = post_params or default_post_params()
        ticket = C.c_uint64(0)
        _check(self.lib, self.ctx, self.lib.vpt_postprocess_device(self.ctx, C.byref(pp), rgba8_device, C.byref(ticket)), "vpt_postprocess_device")
        return int(ticket.value)

    def wait(self, ticket=0):
        """vpt_wait.  ticket 0: everything enqueued so far; otherwise the work up to that ticket (a host that runs one frame ahead of the
        device waits for the PREVIOUS frame's post-process ticket)."""
        _check(self.lib, self.ctx, self.lib.vpt_wait(self.ctx, ticket), "vpt_wait")

    def lab_set(self, key, value):
        """Measurement hooks of include/vpt_lab.h (VPT_LAB_*; laboratory build only): scheduling only, images never depend on them."""
        if not self.lib.has_lab:
            raise VptError("vpt_lab_set: the product library has no laboratory entry points (run with VPT_LAB=1)")
        _check(self.lib, self.ctx, self.lib.vpt_lab_set(self.ctx, key, value), "vpt_lab_set")

    def output_device(self):
        """GetOutputImageView(): device pointer of the RGBA8 image (None before the first post-process)."""
        return self.lib.vpt_output_device(self.ctx)

    def output_to_host(self):
        """vpt_get_output: the RGBA8 image the last post-process left on the device, read back (drains)."""
        out = np.empty((self.height, self.width, 4), np.uint8)
        _check(self.lib, self.ctx, self.lib.vpt_get_output(self.ctx, out.ctypes.data), "vpt_get_output")
        return out

    def radiance(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        _check(self.lib, self.ctx, self.lib.vpt_get_radiance(self.ctx, out.ctypes.data), "vpt_get_radiance")
        return out

    def radiance_to_device(self, ptr):
        _check(self.lib, self.ctx, self.lib.vpt_get_radiance_device(self.ctx, ptr), "vpt_get_radiance_device")

    def set_radiance(self, img, frame_count):
        img = np.ascontiguousarray(img, np.float32)
        assert img.shape == (self.height, self.width, 4)
        _check(self.lib, self.ctx, self.lib.vpt_set_radiance(self.ctx, img.ctypes.data, frame_count), "vpt_set_radiance")

    def shard_floats(self):
        return int(self.lib.vpt_shard_floats(self.ctx))

    def shard_to_device(self, ptr):
        _check(self.lib, self.ctx, self.lib.vpt_get_shard_device(self.ctx, ptr), "vpt_get_shard_device")

    def assemble_shards(self, gathered_ptr, shard_count):
        _check(self.lib, self.ctx, self.lib.vpt_assemble_shards(self.ctx, gathered_ptr, shard_count), "vpt_assemble_shards")

    # ---- PostProcessor API
    def postprocess(self, post_params=None, want_bloom=False):
        pp = post_params or default_post_params()
        out = np.empty((self.height, self.width, 4), np.uint8)
        bloom = np.empty((self.height, self.width, 4), np.float32) if want_bloom else None
        _check(self.lib, self.ctx, self.lib.vpt_postprocess(self.ctx, C.byref(pp), out.ctypes.data, bloom.ctypes.data if want_bloom else None), "vpt_postprocess")
        return (out, bloom) if want_bloom else out

    def stats(self):
        s = _abi.Stats()
        _check(self.lib, self.ctx, self.lib.vpt_get_stats(self.ctx, C.byref(s)), "vpt_get_stats")
        d = {k: getattr(s, k) for k, _ in _abi.Stats._fields_ if k not in ("kernel_launches", "kernel_ms", "stack_spills")}
        d["stack_spills"] = [int(s.stack_spills[0]), int(s.stack_spills[1])]
        d["kernel_launches"] = {n: int(s.kernel_launches[i]) for i, n in enumerate(_abi.KERNEL_NAMES)}
        d["kernel_ms"] = {n: float(s.kernel_ms[i]) for i, n in enumerate(_abi.KERNEL_NAMES)}
        return d

    def reset_stats(self):
        _check(self.lib, self.ctx, self.lib.vpt_reset_stats(self.ctx), "vpt_reset_stats")

    def trace_rays(self, rays):
        rays = np.ascontiguousarray(rays, np.float32)
        hits = np.zeros(len(rays), HIT_DTYPE)
        _check(self.lib, self.ctx, self.lib.vpt_trace_rays(self.ctx, rays.ctypes.data, len(rays), hits.ctypes.data), "vpt_trace_rays")
        return hits


HIT_DTYPE = np.dtype([("t", "<f4"), ("u", "<f4"), ("v", "<f4"), ("primitive", "<u4"), ("instance", "<u4")])
