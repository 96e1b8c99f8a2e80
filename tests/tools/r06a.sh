cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06a
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r06a/gputest.log 2>&1; echo "gputest rc $?"
tail -5 gpurun_out/r06a/gputest.log
timeout 400 python tests/tools/frame_latency.py > gpurun_out/r06a/frame_latency.log 2>&1; cat gpurun_out/r06a/frame_latency.log
PIPE=0 SCENES=cornell FRAMES=0 timeout 600 bash tests/tools/ab_variants.sh "product diag_nolight" 2 > gpurun_out/r06a/ab_nolight.log 2>&1; cat gpurun_out/r06a/ab_nolight.log
SCENES="atrium bust" FRAMES=0 timeout 600 bash tests/tools/ab_variants.sh "product" 1 > gpurun_out/r06a/base_streams.log 2>&1; cat gpurun_out/r06a/base_streams.log
