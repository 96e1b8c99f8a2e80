#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the bench command, then separate PMC
# passes (FETCH_SIZE / WRITE_SIZE cannot share a pass: TCC has 4 slots, MI355X_MICROARCH.md §rocprofv3 PMC slots).
# Outputs land in gpurun_out/prof_*; profiles/summarize.py turns them into the committed summaries.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
BENCH="python bench.py --steps 16 --warmup 2 --no-cpu-baseline"
rm -rf $OUT/prof_kt $OUT/prof_fetch $OUT/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kt -o kt -- $BENCH > $OUT/prof_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_fetch -o fetch -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_write -o write -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/prof_write.log 2>&1
find $OUT/prof_kt $OUT/prof_fetch $OUT/prof_write -type f | head -50
tail -2 $OUT/prof_kt.log
