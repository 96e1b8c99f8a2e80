#!/bin/bash
# rocprofv3 passes over the headline bench (Cornell 1080p depth 8, fused pipeline): kernel trace + separate PMC passes.
# Output under gpurun_out/prof_bench/ ; profiles/summarize_bench_r06.py turns it into profiles/r06_cornell_* and traffic.json.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
BENCH="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-workloads --no-latency --no-live-pmc"
timeout -k 30 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- $BENCH > $OUT/kt.log 2>&1
timeout -k 30 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- $BENCH > $OUT/fetch.log 2>&1
timeout -k 30 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- $BENCH > $OUT/write.log 2>&1
timeout -k 30 900 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/sq -o sq -- $BENCH > $OUT/sq.log 2>&1
timeout -k 30 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/tcc -o tcc -- $BENCH > $OUT/tcc.log 2>&1
# round 4: the fp32 operation mix of the fused kernels (SURVEY 8d secondary roofline: achieved FLOP/s against 157 TF), measured instead of derived
timeout -k 30 900 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/flops -o flops -- $BENCH > $OUT/flops.log 2>&1
# round 4: the LDS side of the whole-path launch (its tree, triangles, stacks and hit ring all live there)
timeout -k 30 900 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/lds -o lds -- $BENCH > $OUT/lds.log 2>&1
find $OUT -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete
grep -h '"value"' $OUT/kt.log | head -c 300
