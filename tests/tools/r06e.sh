cd $GRAFT_REPO_ROOT
bash profiles/collect_bench_r06.sh > gpurun_out/collect_bench.log 2>&1; tail -2 gpurun_out/collect_bench.log
bash profiles/collect_r06.sh atrium 2 > gpurun_out/collect_atrium.log 2>&1; tail -3 gpurun_out/collect_atrium.log
bash profiles/collect_r06.sh bust 2 > gpurun_out/collect_bust.log 2>&1; tail -3 gpurun_out/collect_bust.log
bash profiles/collect_r06.sh atrium 2 3840x2160 > gpurun_out/collect_atrium4k.log 2>&1; tail -3 gpurun_out/collect_atrium4k.log
du -sh gpurun_out
