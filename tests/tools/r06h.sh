cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06h
SCENES="atrium bust" FRAMES=0 timeout 900 bash tests/tools/ab_variants.sh "product scache" 3 > gpurun_out/r06h/ab_scache.log 2>&1; cat gpurun_out/r06h/ab_scache.log
cp vulkan-path-tracer_amd/libvpt_hip.so /tmp/product.so; cp variants/scache.so vulkan-path-tracer_amd/libvpt_hip.so
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_regen.py tests/test_gpu_shadow_modes.py tests/test_gpu_volumes.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python tests/test_gpu_fp32_device.py > /dev/null 2>&1
cp /tmp/product.so vulkan-path-tracer_amd/libvpt_hip.so
timeout 600 python -m pytest tests/test_gpu_fp32_device.py -m gpu -q 2>&1 | tail -15
