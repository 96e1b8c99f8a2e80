cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06b
for s in atrium bust; do
  LAB_QUICK=1 LAB_SPLIT4=1 timeout 900 python tests/tools/trace_lab.py $s 2 > gpurun_out/r06b/trace_lab_split4_$s.json 2> gpurun_out/r06b/trace_lab_split4_$s.log; echo "$s rc $?"
  grep -c '"equal_to_reference": true' gpurun_out/r06b/trace_lab_split4_$s.json; grep -c '"equal_to_reference": false' gpurun_out/r06b/trace_lab_split4_$s.json
  python - <<PY
import json
d=json.load(open("gpurun_out/r06b/trace_lab_split4_$s.json"))
for r in d: print(r["set"], r["variant"], r["ms"], r.get("nodes_per_ray"), r.get("tris_per_ray"), r["equal_to_reference"])
PY
done
