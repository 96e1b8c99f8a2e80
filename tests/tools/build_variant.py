"""Builds variants/<name>.so: the product library with extra -D definitions (A/B builds for tests/tools/ab_variants.sh; variants/ is git-ignored
but travels to the GPU box).  Objects that do not see the definition are shared with the product build.
    python tests/tools/build_variant.py <name> [-DVPT_X=1 ...] [--sources vpt_api.hip,kernels_path.hip]   (default: every source is recompiled)"""
import importlib, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
B = importlib.import_module("vulkan-path-tracer_amd._build")
name = sys.argv[1]
defs = [a for a in sys.argv[2:] if a.startswith("-D")]
only = None
if "--sources" in sys.argv:
    only = sys.argv[sys.argv.index("--sources") + 1].split(",")
B.build(lab=False)   # the product objects
prod = os.path.join(B.HERE, "build", "product")
objdir = os.path.join(B.HERE, "build", "variant_" + name)
os.makedirs(objdir, exist_ok=True)
procs, objs = [], []
for src in B.SOURCES:
    if only is not None and src not in only:
        objs.append(os.path.join(prod, src + ".o")); continue
    obj = os.path.join(objdir, src + ".o"); objs.append(obj)
    cmd = [B.hipcc()] + B.FLAGS + ["-DVPT_LAB=0"] + defs + B.EXTRA_FLAGS.get(src, []) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(B.CSRC, src), "-o", obj]
    procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
for src, p in procs:
    out = p.communicate()[0].decode()
    if p.returncode != 0:
        raise SystemExit("hipcc failed on %s:\n%s" % (src, out))
os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
lib = os.path.join(ROOT, "variants", name + ".so")
subprocess.check_call([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--strip-all"])
open(lib + ".id", "w").write(B.source_id(["-DVPT_LAB=0"] + defs) + "\n")   # not the product's id: counters collected on this build are never attributed to the product
print(lib)
