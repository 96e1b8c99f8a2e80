#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
rm -rf $OUT/prof_a0 $OUT/prof_a1 $OUT/prof_a2 $OUT/prof_a3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_a0 -o a0 -- python tests/gpu_atrium_run.py > $OUT/prof_a0.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/prof_a1 -o a1 -- python tests/gpu_atrium_run.py > $OUT/prof_a1.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_a2 -o a2 -- python tests/gpu_atrium_run.py > $OUT/prof_a2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_a3 -o a3 -- python tests/gpu_atrium_run.py > $OUT/prof_a3.log 2>&1
grep Msamples $OUT/prof_a1.log $OUT/prof_a2.log
