cd $GRAFT_REPO_ROOT
timeout 1200 bash profiles/collect_bench_r06.sh > gpurun_out/collect_bench.log 2>&1; tail -c 300 gpurun_out/collect_bench.log
