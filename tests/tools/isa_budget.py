"""ISA budget of the vote-scheduled traversal kernels (not a pytest; needs only hipcc, no GPU).

Compiles csrc/kernels_trace.hip for gfx950 exactly as _build.py does (wave.hpp VPT_MARK: comment-only
inline asm, present in the product build too, at the head of the vote, of each step kind and of the fetch step), cuts the named kernels out of the assembly, rebuilds
their control-flow graph from labels, fall-throughs and branches, and attributes every instruction to the marker that was passed
last on the way to it.  A step's figure is therefore the number of instructions a wave ISSUES for one step when its lanes take
every side of the step's internal branches (which is what 64 divergent rays do), including the register copies at the loop's back
edge and the vote itself ("vote" region).

    python tests/tools/isa_budget.py [--src DIR] [--json out.json]      # DIR defaults to vulkan-path-tracer_amd/csrc

--src lets the same accounting run over another checkout of csrc/ (e.g. the previous round's, with the markers patched in).
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
_build = __import__("importlib").import_module("vulkan-path-tracer_amd._build")
# exactly the product's flags for this file (VPT_NO_FILE_FLAGS=1 drops the per-file additions, VPT_EXTRA_FLAGS appends: for A/B counts)
FLAGS = _build.FLAGS + ([] if os.environ.get("VPT_NO_FILE_FLAGS") else _build.EXTRA_FLAGS.get("kernels_trace.hip", [])) + os.environ.get("VPT_EXTRA_FLAGS", "").split()

KERNELS = {  # label -> (source file, regex on the mangled name)
    "k_trace_vote<closest> (extend)": ("kernels_trace.hip", r"_ZN3vpt12k_trace_voteILb0ELb0ELb0E(?:Lb1E(?:Lb0E){0,3}(?:Lb1E)?(?:Lb0E)?)?EEv"),
    "k_trace_vote<closest, lab parameters>": ("kernels_trace.hip", r"_ZN3vpt12k_trace_voteILb0ELb0ELb0ELb0E(?:Lb0E){0,5}EEv"),
    "k_trace_shadow<sky>": ("kernels_trace.hip", r"_ZN3vpt14k_trace_shadowILb0ELb0E(?:Lb1E(?:Lb0E(?:Lb1E)?)?)?EEv"),
    "k_trace_shadow<light>": ("kernels_trace.hip", r"_ZN3vpt14k_trace_shadowILb1ELb0E(?:Lb1E(?:Lb0E(?:Lb1E)?)?)?EEv"),
}


def classify(op):
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_call")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_sleep", "s_barrier")):
        return "wait"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("v_"):
        return "valu"
    return "other"


def kernel_text(asm, pattern):
    out, on = [], False
    rx = re.compile("^(" + pattern + r"\S*):")
    for line in asm.splitlines():
        if not on and rx.match(line):
            on = True
            continue
        if on:
            out.append(line)
            if line.strip().startswith("s_endpgm"):
                break
    return out


def parse_blocks(lines):
    """-> list of blocks {label, items: [("ins", op, text) | ("mark", name)], succ: [indices]}"""
    blocks = [{"label": "entry", "items": []}]
    for raw in lines:
        line = raw.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            blocks.append({"label": m.group(1), "items": []})
            continue
        if re.match(r"^; %bb\.\d+:", line):
            blocks.append({"label": None, "items": []})
            continue
        m = re.match(r"^; VPT_MARK (\w+)", line)
        if m:
            blocks[-1]["items"].append(("mark", m.group(1)))
            continue
        code = line.split(";")[0].strip()
        if not code or code.startswith(".") or code.endswith(":"):
            continue
        blocks[-1]["items"].append(("ins", code.split()[0], code))
    index = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
    for i, b in enumerate(blocks):
        succ, falls = [], True
        for kind, *rest in b["items"]:
            if kind != "ins":
                continue
            op, text = rest
            if op.startswith("s_cbranch") or op == "s_branch":
                succ.append(index[text.split()[-1]])
                if op == "s_branch":
                    falls = False
            if op == "s_endpgm":
                falls = False
        if falls and i + 1 < len(blocks):
            succ.append(i + 1)
        b["succ"] = succ
    return blocks


def region_instructions(blocks):
    """region -> set of (block, item) issued between passing that region's marker and passing the next marker of any kind.
    A block at which two regions join (the loop's back edge) counts for both: each kind of step issues it once."""
    starts = collections.defaultdict(list)
    for bi, b in enumerate(blocks):
        for k, it in enumerate(b["items"]):
            if it[0] == "mark":
                starts[it[1]].append((bi, k + 1))
    starts["prologue"].append((0, 0))
    regions = {}
    for region, entry_points in starts.items():
        got, seen, work = set(), set(), list(entry_points)
        while work:
            bi, k = work.pop()
            if (bi, k) in seen:
                continue
            seen.add((bi, k))
            items = blocks[bi]["items"]
            stopped = False
            while k < len(items):
                if items[k][0] == "mark":
                    stopped = True
                    break
                got.add((bi, k))
                k += 1
            if not stopped:
                for s2 in blocks[bi]["succ"]:
                    work.append((s2, 0))
        regions[region] = got
    # "vote>X": the part of the vote region that lies on a way to marker X (forward-reachable from the vote marker and
    # backward-reachable from X without crossing a marker) = what one iteration issues BEFORE it enters step X
    pred = collections.defaultdict(list)
    for bi, b in enumerate(blocks):
        for s2 in b["succ"]:
            pred[s2].append(bi)
    vote = regions.get("vote", set())
    for target, entry_points in starts.items():
        if target in ("vote", "prologue"):
            continue
        back, seen, work = set(), set(), [(bi, k - 2) for bi, k in entry_points]   # item before the marker
        while work:
            bi, k = work.pop()
            if (bi, k) in seen:
                continue
            seen.add((bi, k))
            items = blocks[bi]["items"]
            stopped = False
            while k >= 0:
                if items[k][0] == "mark":
                    stopped = True
                    break
                back.add((bi, k))
                k -= 1
            if not stopped:
                for p2 in pred[bi]:
                    work.append((p2, len(blocks[p2]["items"]) - 1))
        regions["vote>" + target] = vote & back
    return regions


def budget(blocks):
    regions = region_instructions(blocks)
    table = collections.defaultdict(lambda: collections.Counter())
    detail = collections.defaultdict(lambda: collections.Counter())
    for region, members in regions.items():
      for (bi, k) in members:
        _, op, text = blocks[bi]["items"][k]
        cls = classify(op)
        table[region][cls] += 1
        if cls == "valu":
            base = op.replace("_e32", "").replace("_e64", "")
            group = ("v_mov/copy" if base.startswith(("v_mov", "v_pk_mov", "v_accvgpr")) else
                     "v_cndmask" if base.startswith("v_cndmask") else
                     "v_cmp" if base.startswith("v_cmp") else
                     "v_cvt_f32_ubyte" if base.startswith("v_cvt_f32_ubyte") else
                     "fma/mul/add" if base.startswith(("v_fma", "v_fmac", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_pk_", "v_mad_f32")) else
                     "min/max" if base.startswith(("v_min", "v_max", "v_med")) else
                     "div/rcp/sqrt" if base.startswith(("v_div", "v_rcp", "v_sqrt", "v_rsq")) else
                     "int/addr" if base.startswith(("v_lshl", "v_lshr", "v_and", "v_or", "v_add_u", "v_sub_u", "v_mad_u", "v_add_co", "v_not", "v_bfe", "v_bfi",
                                                    "v_mbcnt", "v_readfirstlane", "v_readlane", "v_writelane", "v_mul_u", "v_mul_lo", "v_mul_hi", "v_ashr", "v_xor", "v_sub_co", "v_addc", "v_add3", "v_xad", "v_perm", "v_alignbit")) else
                     "other")
            detail[region][group] += 1
    return table, detail, []


def compile_asm(src_dir, name, tmp):
    out = os.path.join(tmp, name + ".s")
    cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ["-I", src_dir, "--cuda-device-only", "-S", os.path.join(src_dir, name), "-o", out]
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def main():
    src = os.path.join(ROOT, "vulkan-path-tracer_amd", "csrc")
    out_json = None
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--src":
            src = os.path.abspath(args.pop(0))
        elif a == "--json":
            out_json = args.pop(0)
    result = {}
    with tempfile.TemporaryDirectory() as tmp:
        asm = {}
        for label, (fname, pattern) in KERNELS.items():
            lines = []
            for cand in (fname, "kernels_stream.hip"):   # round 2 kept k_trace_shadow in kernels_stream.hip
                if cand not in asm:
                    asm[cand] = compile_asm(src, cand, tmp)
                lines = kernel_text(asm[cand], pattern)
                if lines:
                    fname = cand
                    break
            if not lines:
                print("%s: kernel not found" % label, file=sys.stderr)
                continue
            blocks = parse_blocks(lines)
            table, detail, conflicts = budget(blocks)
            meta = {}
            # the kernel's entry in the .amdhsa metadata: the keys of one kernel's map are sorted, so .name comes between them
            entries = re.split(r"\n  - \.", asm[fname])
            for e in entries:
                if re.search(r"\.name:\s+" + pattern, e):
                    for key in ("vgpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count"):
                        mm = re.search(r"\." + key + r":\s+(\d+)", e)
                        if mm:
                            meta[key] = int(mm.group(1))
            result[label] = {"regions": {r: dict(c) for r, c in table.items()}, "valu_detail": {r: dict(c) for r, c in detail.items()},
                             "conflicts": conflicts, "meta": meta}
            print("## %s   %s" % (label, meta))
            print("| region | VALU | SALU | VMEM | LDS | SMEM | branch | wait/nop | VALU by kind |")
            print("|---|---|---|---|---|---|---|---|---|")
            for region in ("vote>node", "node", "vote>tri", "tri", "vote>exit", "vote>fetch", "fetch", "exit", "done", "prologue", "vote"):
                c = table.get(region)
                if not c:
                    continue
                kinds = ", ".join("%s %d" % kv for kv in sorted(detail[region].items(), key=lambda kv: -kv[1]))
                print("| %s | %d | %d | %d | %d | %d | %d | %d | %s |" % (region, c["valu"], c["salu"], c["vmem"], c["lds"], c["smem"], c["branch"], c["wait"], kinds))
            if conflicts:
                print("attribution conflicts (block, first, other):", conflicts)
            print()
    if out_json:
        with open(out_json, "w") as f:
            json.dump(result, f, indent=1)


if __name__ == "__main__":
    main()
