cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06g
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r06g/gputest_product.log 2>&1; echo "gputest product rc $?"; tail -4 gpurun_out/r06g/gputest_product.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/r06g/bench.json 2> gpurun_out/r06g/bench.err; echo "bench rc $?"; head -c 600 gpurun_out/r06g/bench.json; echo
timeout 900 python bench.py > gpurun_out/r06g/bench_run2.json 2> gpurun_out/r06g/bench_run2.err; echo "bench2 rc $?"; head -c 300 gpurun_out/r06g/bench_run2.json; echo
timeout 300 python tests/tools/shard_rate.py > gpurun_out/r06g/shard_rate.log 2>&1; cat gpurun_out/r06g/shard_rate.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06g/smoke.log 2>&1; echo "smoke rc $?"; tail -3 gpurun_out/r06g/smoke.log
VPT_LAB=1 timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bench.py --deselect tests/test_gpu_comm.py > gpurun_out/r06g/gputest_lab.log 2>&1; echo "gputest lab rc $?"; tail -3 gpurun_out/r06g/gputest_lab.log | cut -c1-300
