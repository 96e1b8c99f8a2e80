"""Static instruction budget of the shade stage (not a pytest; needs only hipcc, no GPU).

1. Compiles csrc/kernels_stream.hip for gfx950 with the product's flags plus line tables, cuts k_shade_stream<kShadeAny> out of the assembly and
   attributes every VALU instruction to the source line it was generated for (the innermost inlined function), summed per source function.
2. Compiles one tiny kernel per contract / shading function and counts its instructions: what one call costs.

    python tests/tools/shade_budget.py [--json out.json]
    python tests/tools/shade_budget.py --fused        # the same breakdown for the fused per-bounce kernels (kernels_path.hip)
    python tests/tools/shade_budget.py --whole        # ... and for the whole-path launch k_whole
"""
import collections, importlib, json, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
_build = importlib.import_module("vulkan-path-tracer_amd._build")
CSRC = os.path.join(ROOT, "vulkan-path-tracer_amd", "csrc")
FLAGS = [f for f in _build.FLAGS if f not in ("-fPIC",)] + _build.EXTRA_FLAGS.get("kernels_stream.hip", [])   # (kernels_path.hip has the same per-file flags)

def asm(src, extra=()):
    out = tempfile.mktemp(suffix=".s")
    cmd = ["hipcc", "-S", "--cuda-device-only", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "-o", out, src] + FLAGS + list(extra)
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(out).read(); os.unlink(out)
    return text.splitlines()

def function_ranges(path):
    """[(first line, last line, name)] of the top-level / member functions of a source file, by a brace count good enough for this code base."""
    rows, depth, cur = [], 0, None
    rx = re.compile(r"(?:VPT_HD|__device__|__global__)[^;{]*?\b(\w+)\s*\([^;]*\)\s*(?:const\s*)?\{")
    for n, line in enumerate(open(path), 1):
        code = line.split("//")[0]
        if cur is None:
            m = rx.search(code)
            if m: cur = [n, n, m.group(1)]; depth = 0
        if cur is not None:
            depth += code.count("{") - code.count("}")
            cur[1] = n
            if depth <= 0 and "{" in "".join(open(path).readlines()[cur[0] - 1:n]):
                rows.append(tuple(cur)); cur = None
    return rows

def shade_kernel_by_function(src="kernels_stream.hip", symbol="_ZN3vpt14k_shade_streamILin1EEE"):
    lines = asm(os.path.join(CSRC, src), ["-gline-tables-only"])
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m: files[int(m.group(1))] = os.path.basename(m.group(3) or m.group(2))
    start = next(i for i, l in enumerate(lines) if l.startswith(symbol) and "@" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    by_line, cur, total = collections.Counter(), None, collections.Counter()
    for l in lines[start + 1:end]:
        t = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m: cur = (files.get(int(m.group(1)), "?"), int(m.group(2))); continue
        if not t or t.startswith((";", ".")) or t.endswith(":"): continue
        op = t.split()[0]
        kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "mem"
        total[kind] += 1
        if kind == "valu": by_line[cur] += 1
    ranges = {}
    for f in ("vpt_fp32.h", "shading.hpp", "shade_core.hpp", "kernels_stream.hip", "kernels_path.hip", "traverse.hpp", "vote.hpp", "wave.hpp", "volume.hpp", "atmosphere.hpp"):
        p = os.path.join(ROOT, "include", f) if f == "vpt_fp32.h" else os.path.join(CSRC, f)
        ranges[f] = function_ranges(p)
    by_fn = collections.Counter()
    for (f, ln), c in by_line.items():
        name = next((r[2] for r in ranges.get(f, []) if r[0] <= ln <= r[1]), "(other)")
        by_fn[(f, name)] += c
    return total, by_fn

MICRO = r'''
#include <hip/hip_runtime.h>
#include "shade_core.hpp"
using namespace vpt; using namespace vptfp;
#define K1(name, expr) extern "C" __global__ void k_##name(float* p) { float x = p[threadIdx.x], y = p[threadIdx.x + 64]; (void)y; p[threadIdx.x] = (expr); }
K1(empty, x) K1(sin, sin_(x)) K1(cos, cos_(x)) K1(acos, acos_(x)) K1(asin, asin_(x)) K1(atan2, atan2_(x, y)) K1(pow, pow_(x, y)) K1(log, log_(x)) K1(exp, exp_(x)) K1(sqrt, sqrt_(x)) K1(div, x / y)
extern "C" __global__ void k_sincos(float* p) { float s, c; sincos_(p[threadIdx.x], &s, &c); p[threadIdx.x] = s + c; }
extern "C" __global__ void k_normalize(float* p) { V3 v = normalize(v3(p[threadIdx.x], p[64 + threadIdx.x], p[128 + threadIdx.x])); p[threadIdx.x] = v.x + v.y + v.z; }
extern "C" __global__ void k_uniform_draw(float* p) { Rng r; r.s = __float_as_uint(p[threadIdx.x]); float a = r.uf(); p[threadIdx.x] = a + __uint_as_float(r.s); }
extern "C" __global__ void k_ggx_sample(float* p) { Rng r; r.s = __float_as_uint(p[threadIdx.x]); V3 h = ggx_sample(r, v3(p[64 + threadIdx.x], p[128 + threadIdx.x], p[192 + threadIdx.x]), p[256], p[257]); p[threadIdx.x] = h.x + h.y + h.z + __uint_as_float(r.s); }
extern "C" __global__ void k_bsdf_eval(Bsdf* bp, float* p) { Bsdf b = *bp; V3 V = v3(p[threadIdx.x], p[64 + threadIdx.x], p[128 + threadIdx.x]), L = v3(p[192 + threadIdx.x], p[256 + threadIdx.x], p[320 + threadIdx.x]); Eval e = b.eval(V, L, p[400], p[401], p[402]); p[threadIdx.x] = e.f.x + e.f.y + e.f.z + e.pdf; }
extern "C" __global__ void k_texel_coords(float* p) { int a, b; float w; texel_coords(p[threadIdx.x], 1024, true, &a, &b, &w); p[threadIdx.x] = w + (float)(a + b); }
extern "C" __global__ void k_rotate(float* p) { V3 v = rotate_sc(v3(p[threadIdx.x], p[64 + threadIdx.x], p[128 + threadIdx.x]), v3(0.0f, 1.0f, 0.0f), p[300], p[301]); p[threadIdx.x] = v.x + v.y + v.z; }
extern "C" __global__ void k_world_to_tangent(float* p) { SurfaceFrame s; s.T = v3(p[0], p[1], p[2]); s.B = v3(p[3], p[4], p[5]); s.N = v3(p[6], p[7], p[8]); V3 v = s.world_to_tangent(v3(p[64 + threadIdx.x], p[128 + threadIdx.x], p[192 + threadIdx.x])); p[threadIdx.x] = v.x + v.y + v.z; }
extern "C" __global__ void k_bilinear_rgba8(const uint8_t* tx, TexDesc* d, float* p) { TexTaps k; tex_issue(tx, *d, p[threadIdx.x], p[64 + threadIdx.x], k); V4 v = tex_finish(k); p[threadIdx.x] = v.x + v.y + v.z + v.w; }
'''

def micro_costs():
    src = tempfile.mktemp(suffix=".hip"); open(src, "w").write(MICRO)
    lines = asm(src); os.unlink(src)
    out, i = {}, 0
    while i < len(lines):
        m = re.match(r"^(k_\w+):", lines[i])
        if m:
            ops, j = collections.Counter(), i + 1
            while not lines[j].startswith(".Lfunc_end"):
                t = lines[j].strip()
                if t and not t.startswith((";", ".")) and not t.endswith(":"): ops[t.split()[0]] += 1
                j += 1
            out[m.group(1)[2:]] = {"valu": sum(v for k, v in ops.items() if k.startswith("v_")),
                                   "transcendental": sum(v for k, v in ops.items() if re.match(r"v_(rcp|sqrt|rsq|exp|log|sin|cos)_", k)),
                                   "ieee_div_sequences": ops["v_div_fixup_f32"]}
            i = j
        i += 1
    base = out.pop("empty")["valu"]
    for v in out.values(): v["valu"] = max(0, v["valu"] - base)   # minus the load / store scaffolding
    return out

if __name__ == "__main__":
    if "--fused" in sys.argv:   # the headline's two kernels: k_bounce<LDS scene> for bounces >= 1 and its FIRST variant (camera ray + bounce 0)
        for label, sym in (("k_bounce<LDS, bounce >= 1>", "_ZN3vpt8k_bounceILb1ELb0ELb0ELb0ELb0EEE"), ("k_bounce<LDS, FIRST>", "_ZN3vpt8k_bounceILb1ELb0ELb1ELb0ELb0EEE")):
            total, by_fn = shade_kernel_by_function("kernels_path.hip", sym)
            print("%s: %d VALU, %d SALU, %d memory instructions (static)\n" % (label, total["valu"], total["salu"], total["mem"]))
            print("| source function (innermost inlined) | VALU instructions | share |\n|---|---|---|")
            for (f, n), c in by_fn.most_common(16):
                print("| `%s` (%s) | %d | %.1f %% |" % (n, f, c, 100.0 * c / total["valu"]))
            print()
        sys.exit(0)
    if "--whole" in sys.argv:   # the whole-path launch (round 4): scene-class and general instantiation
        for label, sym in (("k_whole<PLAIN>", "_ZN3vpt7k_wholeILb0ELb0ELb1EEE"), ("k_whole<general>", "_ZN3vpt7k_wholeILb0ELb0ELb0EEE")):
            total, by_fn = shade_kernel_by_function("kernels_path.hip", sym)
            print("%s: %d VALU, %d SALU, %d memory instructions (static)\n" % (label, total["valu"], total["salu"], total["mem"]))
            print("| source function (innermost inlined) | VALU instructions | share |\n|---|---|---|")
            for (f, n), c in by_fn.most_common(24):
                print("| `%s` (%s) | %d | %.1f %% |" % (n, f, c, 100.0 * c / total["valu"]))
            print()
        sys.exit(0)
    total, by_fn = shade_kernel_by_function()
    micro = micro_costs()
    print("k_shade_stream<kShadeAny>: %d VALU, %d SALU, %d memory instructions (static)\n" % (total["valu"], total["salu"], total["mem"]))
    print("| source function (innermost inlined) | VALU instructions | share |\n|---|---|---|")
    for (f, n), c in by_fn.most_common(28):
        print("| `%s` (%s) | %d | %.1f %% |" % (n, f, c, 100.0 * c / total["valu"]))
    print("\n| one call of | VALU | of which transcendental | IEEE division sequences |\n|---|---|---|---|")
    for k, v in sorted(micro.items(), key=lambda kv: -kv[1]["valu"]):
        print("| `%s` | %d | %d | %d |" % (k, v["valu"], v["transcendental"], v["ieee_div_sequences"]))
    if "--json" in sys.argv:
        json.dump({"kernel": dict(total), "by_function": {"%s:%s" % k: v for k, v in by_fn.items()}, "per_call": micro}, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
