#!/bin/bash
# rocprofv3 passes over tests/gpu_atrium_run.py for one (SCENE, PIPE[, SIZE]): kernel trace + separate PMC passes (never combined with
# tracing other than --kernel-trace), on the library's OWN batch schedule (FRAMES=0: batches of 904 frames with 113 frames of paths resident at 1080p — what bench.py's workloads run).
# Usage: profiles/collect_r06.sh <scene> <pipe> [WxH] ; output under gpurun_out/prof_<scene>_p<pipe>[_WxH]/   (the fp32-mix and instruction-cache passes are
# collected for the headline bench only, profiles/collect_bench_r05.sh; round 6 adds the L2's memory-side request mix, for the join stage's paragraph in DESIGN.md)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
export SCENE=$1 PIPE=$2 FRAMES=${FRAMES:-0}
if [ -n "${3:-}" ]; then export SIZE=$3; OUT=gpurun_out/prof_${SCENE}_p${PIPE}_$3; else OUT=gpurun_out/prof_${SCENE}_p${PIPE}; fi
[ -n "${KEEP:-}" ] || rm -rf $OUT; mkdir -p $OUT
timeout -k 30 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python tests/gpu_atrium_run.py > $OUT/kt.log 2>&1
timeout -k 30 600 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq1 -o sq1 -- python tests/gpu_atrium_run.py > $OUT/sq1.log 2>&1
timeout -k 30 600 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/sq2 -o sq2 -- python tests/gpu_atrium_run.py > $OUT/sq2.log 2>&1
timeout -k 30 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/tcc -o tcc -- python tests/gpu_atrium_run.py > $OUT/tcc.log 2>&1
timeout -k 30 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python tests/gpu_atrium_run.py > $OUT/fetch.log 2>&1
timeout -k 30 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python tests/gpu_atrium_run.py > $OUT/write.log 2>&1
# the texture-address / vector-L1 pipeline (is a kernel bound by scattered loads? profiles/r04_trace_isa_budget.md)
timeout -k 30 600 rocprofv3 --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/ta -o ta -- python tests/gpu_atrium_run.py > $OUT/ta.log 2>&1
# the L2's memory-side request mix: how many of the requests that leave L2 are 32-byte reads / partial writes (a stage of scattered 16-byte records: k_join)
# (two counters per pass: four TCC sums in one pass exceed what the hardware collects at once — rocprofv3 then aborts and hangs in its signal handler, hence the timeouts)
timeout -k 30 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/tccrd -o tccrd -- python tests/gpu_atrium_run.py > $OUT/tccrd.log 2>&1
timeout -k 30 600 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/tccwr -o tccwr -- python tests/gpu_atrium_run.py > $OUT/tccwr.log 2>&1
grep -h Msamples $OUT/*.log
# keep only what the summaries need (the merged gpurun_out is capped at 64 MiB)
find $OUT -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete
