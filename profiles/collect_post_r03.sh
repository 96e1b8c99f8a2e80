#!/bin/bash
# rocprofv3 passes over tests/gpu_post_run.py for one SIZE (1080p | 4k) and SCHEDULE (fused | reference_passes): kernel trace + separate
# FETCH_SIZE / WRITE_SIZE passes (never combined with tracing other than --kernel-trace).  Output: gpurun_out/prof_post_<size>_<schedule>/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
export SIZE=$1 SCHEDULE=$2 REPS=${REPS:-20}
OUT=gpurun_out/prof_post_${SIZE}_${SCHEDULE}
rm -rf $OUT; mkdir -p $OUT
python tests/gpu_post_run.py > $OUT/events.json 2> $OUT/events.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python tests/gpu_post_run.py > $OUT/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python tests/gpu_post_run.py > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python tests/gpu_post_run.py > $OUT/write.log 2>&1
cat $OUT/events.json
find $OUT -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete
