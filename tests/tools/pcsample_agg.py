"""Aggregates a rocprofv3 PC-sampling CSV (probe helper, not a pytest): sample counts per instruction address, written as JSON."""
import collections, csv, glob, json, sys
src, out = sys.argv[1], sys.argv[2]
files = glob.glob(src + "/**/*pc_sampling*.csv", recursive=True)
print("files", files)
res = {}
for f in files:
    rd = csv.DictReader(open(f))
    print(f, rd.fieldnames)
    cnt = collections.Counter(); n = 0
    for r in rd:
        n += 1
        key = (r.get("Code_Object_Id") or r.get("Instruction_Code_Object_Id") or "", r.get("Code_Object_Offset") or r.get("Instruction_Code_Object_Offset") or r.get("Instruction") or "", r.get("Instruction", ""), r.get("Instruction_Comment", ""))
        cnt[key] += 1
    res[f.split("/")[-1]] = {"samples": n, "top": [[list(k), v] for k, v in cnt.most_common(4000)]}
json.dump(res, open(out, "w"))
