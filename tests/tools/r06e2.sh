cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/prof_atrium_p2; mkdir -p $OUT
export SCENE=atrium PIPE=2 FRAMES=0
timeout -k 30 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/tccrd -o tccrd -- python tests/gpu_atrium_run.py > $OUT/tccrd.log 2>&1; echo "tccrd rc $?"
timeout -k 30 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $OUT/tccwr -o tccwr -- python tests/gpu_atrium_run.py > $OUT/tccwr.log 2>&1; echo "tccwr rc $?"
find $OUT -name "*.csv" ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -delete
tail -3 $OUT/tccrd.log | cut -c1-200
timeout 900 bash profiles/collect_r06.sh bust 2 > gpurun_out/collect_bust.log 2>&1; tail -3 gpurun_out/collect_bust.log | cut -c1-200
timeout 900 bash profiles/collect_r06.sh atrium 2 3840x2160 > gpurun_out/collect_atrium4k.log 2>&1; tail -3 gpurun_out/collect_atrium4k.log | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_fp32_device.py -m gpu -x -q 2>&1 | tail -5
