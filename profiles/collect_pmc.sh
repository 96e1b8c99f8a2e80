#!/bin/bash
# SQ-side counters of the path kernels (own pass: SQ has 8 slots). Output: gpurun_out/prof_sq/
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
rm -rf $OUT/prof_sq $OUT/prof_sq2
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --output-format csv -d $OUT/prof_sq -o sq -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_sq.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_sq2 -o sq2 -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/prof_sq2.log 2>&1
tail -3 $OUT/prof_sq.log | cut -c1-300
tail -3 $OUT/prof_sq2.log | cut -c1-300
