"""Register / scratch report of the product's kernels from the compiler's own output (not a pytest; needs hipcc only, no GPU): every kernel file is compiled to
gfx950 assembly with the product's flags plus line tables, and for every kernel the report lists VGPRs, SGPRs, occupancy, scratch bytes, static VALU / SALU counts,
scalar-spill lane operations (v_writelane / v_readlane: scalar registers that live in VGPR lanes) and, for every `scratch_` instruction, the innermost source line
it was generated for — which is how round 5's two scratch sources in the shade stage were found (a selected-address store in sample_emissive; wave-uniform state the
compiler had to treat as divergent).
    python tests/tools/spill_report.py [--md profiles/r05_spill_report.md] [--all]     (default: kernels a BASELINE config launches)"""
import collections, importlib, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
B = importlib.import_module("vulkan-path-tracer_amd._build")
CSRC = os.path.join(ROOT, "vulkan-path-tracer_amd", "csrc")
# the kernels of the BASELINE configs (bench.py's workloads): demangled-name fragments
MAIN = ("k_whole<false, false, true>", "k_whole<false, false, false>", "k_shade_stream<-1>", "k_trace_vote<false, false, false, true, false, false, false, true, false>",
        "k_trace_shadow<true, false, true, false, true>", "k_trace_shadow<false, false, true, false, true>", "k_join", "k_refill_stream", "k_raygen_stream", "k_finish<false, false>", "k_resolve",
        "k_post_final<true, true>", "k_bloom_down<true>", "k_bloom_tail<true>", "k_bloom_up_chain", "k_bloom_down_chain")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def report(src):
    flags = [f for f in B.FLAGS if f != "-fPIC"] + ["-DVPT_LAB=0"] + B.EXTRA_FLAGS.get(src, []) + ["-gline-tables-only"]
    out = tempfile.mktemp(suffix=".s")
    subprocess.run([B.hipcc(), "-S", "--cuda-device-only", "-I" + CSRC, "-I" + os.path.join(ROOT, "include"), "-o", out, os.path.join(CSRC, src)] + flags,
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n"); os.unlink(out)
    starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:\s+; @", l)]
    names = demangle([n for _, n in starts])
    rows = []
    for (i, n), (j, _) in zip(starts, starts[1:] + [(len(lines), "")]):
        body = lines[i:j]
        get = lambda key: next((int(re.search(r"(\d+)", x.split(key)[1]).group(1)) for x in body if key in x), None)
        loc, spills = "", collections.Counter()
        for x in body:
            if ".loc" in x and ";" in x:
                loc = x.split(";")[-1].strip().split(" @[")[0].replace("vulkan-path-tracer_amd/csrc/", "")
            if re.match(r"^\s+scratch_", x):
                spills[("store " if "_store" in x else "load  ") + loc] += 1
        rows.append({"kernel": re.sub(r"\(.*", "", names[n]).replace("void ", "").replace("vpt::", ""), "vgprs": get("; NumVgprs:"), "sgprs": get("; TotalNumSgprs:"), "occupancy": get("; Occupancy:"),
                     "scratch_bytes": get("; ScratchSize:"), "valu": sum(1 for x in body if re.match(r"^\s+v_", x)), "salu": sum(1 for x in body if re.match(r"^\s+s_", x)),
                     "lane_ops": sum(1 for x in body if "v_readlane" in x or "v_writelane" in x), "scratch_instr": sum(spills.values()), "where": spills})
    return rows


if __name__ == "__main__":
    everything = "--all" in sys.argv
    rows = []
    for src in ("kernels_path.hip", "kernels_stream.hip", "kernels_trace.hip", "kernels_post.hip", "kernels_media.hip"):
        rows += [dict(r, file=src) for r in report(src)]
    rows = [r for r in rows if everything or any(r["kernel"] == m or r["kernel"].startswith(m) for m in MAIN)]
    out = ["# Registers and scratch of the shipped kernels, from the compiler's output (`tests/tools/spill_report.py`, source id %s)" % B.source_id(), "",
           "Static counts of the gfx950 assembly (`hipcc -S` with the product's flags + line tables, `-DVPT_LAB=0`).  lane ops = `v_writelane` / `v_readlane`: scalar registers the kernel keeps in VGPR lanes.", "",
           "| kernel | file | VGPRs | SGPRs | waves / SIMD | scratch B / lane | scratch instr | VALU | SALU | scalar-spill lane ops |", "|---|---|---|---|---|---|---|---|---|---|"]
    for r in rows:
        out.append("| `%s` | %s | %s | %s | %s | %s | %d | %d | %d | %d |" % (r["kernel"], r["file"], r["vgprs"], r["sgprs"], r["occupancy"], r["scratch_bytes"], r["scratch_instr"], r["valu"], r["salu"], r["lane_ops"]))
    out += ["", "Where the scratch instructions of those kernels come from (innermost source line):", ""]
    for r in rows:
        if r["where"]:
            out.append("* `%s`: " % r["kernel"] + "; ".join("%d x %s" % (c, w) for w, c in sorted(r["where"].items(), key=lambda kv: -kv[1])[:12]))
    text = "\n".join(out) + "\n"
    if "--md" in sys.argv:
        open(os.path.join(ROOT, sys.argv[sys.argv.index("--md") + 1]), "w").write(text)
    print(text)
