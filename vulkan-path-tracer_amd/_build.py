"""Builds libvpt_hip.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

-ffp-contract=off and no fast-math are part of the parity contract (include/vpt_fp32.h): the only
fused multiply-adds in the binary are the explicit vptfp::fma() calls.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvpt_hip.so")
# The LABORATORY build: the same sources with -DVPT_LAB=1 — every kernel variant that was measured against the product kernels and found
# slower, round 1's stage kernels (VPT_PIPELINE_STAGED_R1) and the vpt_lab_* entry points of include/vpt_lab.h.  The product library has none of it.
LIB_LAB = os.path.join(HERE, "libvpt_hip_lab.so")
SOURCES = ["kernels_path.hip", "kernels_trace.hip", "kernels_stream.hip", "kernels_media.hip", "kernels_post.hip", "kernels_lut.hip", "vpt_api.hip", "bvh_build.cpp"]
HEADERS = ["device_types.hpp", "kernels.hpp", "shading.hpp", "traverse.hpp", "bvh_build.hpp", "volume.hpp", "atmosphere.hpp", "wave.hpp", "shade_core.hpp", "vote.hpp",
           os.path.join("..", "..", "include", "vpt.h"), os.path.join("..", "..", "include", "vpt_lab.h"), os.path.join("..", "..", "include", "vpt_fp32.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-result", "-Wno-pass-failed"]
# Per-file additions.  The traversal kernels are VALU-issue-bound and their triangle test is 51 scalar fp32 operations: the SLP
# vectoriser turns 28 of them into 14 packed ones at the price of 20-30 register moves to pair the operands up
# (profiles/r03_trace_isa_budget.md), a net loss in issued instructions, so it is off for that file.  Values are unaffected:
# packed and scalar fp32 operations round identically and no contraction is allowed either way.
EXTRA_FLAGS = {"kernels_trace.hip": ["-fno-slp-vectorize"], "kernels_stream.hip": ["-fno-slp-vectorize"], "kernels_path.hip": ["-fno-slp-vectorize"], "kernels_media.hip": ["-fno-slp-vectorize"]}


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def source_id(defines=("-DVPT_LAB=0",)):
    """sha256 (16 hex digits) over everything the library is compiled from: kernel sources, headers, this file's flags and the build's -D list
    (the product's is -DVPT_LAB=0; the laboratory build and tests/tools/build_variant.py pass theirs, so a variant never carries the product's id).
    profiles/traffic.json carries the id of the build its rocprofv3 counters were collected on; bench.py copies an entry only when it equals the current one."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES) + sorted(HEADERS):
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(repr((FLAGS, sorted(EXTRA_FLAGS.items()), sorted(defines))).encode())
    return h.hexdigest()[:16]


def needs_build(lab=False):
    lib = LIB_LAB if lab else LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    # (this file holds the compiler flags: a change here rebuilds too)
    return os.path.getmtime(__file__) > t or any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, lab=False):
    lib = LIB_LAB if lab else LIB
    if not force and not needs_build(lab):
        return lib
    import fcntl
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    with open(os.path.join(HERE, "build", ".lock"), "w") as lock:   # one builder at a time (pytest-xdist workers, a bench beside a test run): the others find the library built
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not needs_build(lab):
            return lib
        return _build_locked(lib, force, verbose, lab)


def _build_locked(lib, force, verbose, lab):
    objs = []
    procs = []
    objdir = os.path.join(HERE, "build", "lab" if lab else "product")
    os.makedirs(objdir, exist_ok=True)
    newest_header = max([os.path.getmtime(__file__)] + [os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS])
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(newest_header, os.path.getmtime(os.path.join(CSRC, src))):
            continue   # this object is newer than its source and every header
        cmd = [hipcc()] + FLAGS + ["-DVPT_LAB=%d" % (1 if lab else 0)] + EXTRA_FLAGS.get(src, []) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out))
        if verbose and out.strip():
            print(out)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--strip-all"]
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, lab="--lab" in sys.argv))
