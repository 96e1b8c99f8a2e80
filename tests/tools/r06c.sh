cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06c
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_regen.py tests/test_gpu_volumes.py tests/test_gpu_atmosphere.py -m gpu -x -q 2>&1 | tail -3
SCENES="atrium bust" FRAMES=0 timeout 1500 bash tests/tools/ab_variants.sh "product envrows pairwise" 2 > gpurun_out/r06c/ab_envrows_pairwise.log 2>&1; cat gpurun_out/r06c/ab_envrows_pairwise.log
