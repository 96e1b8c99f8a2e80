#!/bin/bash
# Where do k_shade_stream's HBM-side bytes come from?  FETCH_SIZE / WRITE_SIZE passes over the atrium (1080p, 64-frame batches, all resident) in variants that change
# only what a hit has to FETCH: sky NEE off (CLEAR_FLAGS=1: no alias entry, no environment texels of the sample), light NEE off (2: no light record / light triangle),
# energy-compensation taps off (16), every value texture 1x1 (no texel lines), a 64x32 environment (alias table + texels fit L2).  Output: gpurun_out/shade_sources/<variant>/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp SCENE=atrium PIPE=2 FRAMES=64 RESIDENT=64
OUT=gpurun_out/shade_sources
rm -rf $OUT; mkdir -p $OUT
run() {  # name, env assignments...
  local name=$1; shift
  env "$@" rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$name -o f -- python tests/gpu_atrium_run.py > $OUT/$name.log 2>&1
}
run base X=0
run no_sky_nee CLEAR_FLAGS=1
run no_light_nee CLEAR_FLAGS=2
run no_ec_taps CLEAR_FLAGS=16
run tex1x1 VARIANT=tex1x1
run env64 VARIANT=env64
grep -h Msamples $OUT/*.log
find $OUT -name "*.csv" ! -name "*counter_collection.csv" -delete
