#!/bin/bash
# A/B of library builds inside ONE gpurun call (box-to-box spread is ~5 %): variants/<name>.so are copied over the product library in turn.
#   tests/tools/ab_variants.sh "<name> <name> ..." [rounds]    env: SCENES="atrium bust"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp vulkan-path-tracer_amd/libvpt_hip.so /tmp/product.so
for r in $(seq 1 ${2:-2}); do
  for v in $1; do
    cp variants/$v.so vulkan-path-tracer_amd/libvpt_hip.so
    for s in ${SCENES:-atrium}; do
      echo "== $v $s round $r: $(SCENE=$s FRAMES=${FRAMES:-129} PROFILE=1 python tests/gpu_atrium_run.py 2>&1 | tail -1)"
    done
  done
done
cp /tmp/product.so vulkan-path-tracer_amd/libvpt_hip.so
