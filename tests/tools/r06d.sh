cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06d
cp vulkan-path-tracer_amd/libvpt_hip.so /tmp/product.so
# per-frame latency under the finisher variants (alternating, one box)
for r in 1 2; do
  for v in product fa1 fa2 fa4 fb2; do
    cp variants/$v.so vulkan-path-tracer_amd/libvpt_hip.so
    echo "== $v round $r"; timeout 300 python tests/tools/frame_latency.py atrium,bust 30 2>&1 | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['scene'], d['blocking_frame_ms'], d['async_1_in_flight_ms'], d['async_2_in_flight_ms'], d['async_3_in_flight_ms'])"
  done
done > gpurun_out/r06d/finish_latency_ab.log 2>&1
cp /tmp/product.so vulkan-path-tracer_amd/libvpt_hip.so
cat gpurun_out/r06d/finish_latency_ab.log
# in-batch throughput with a higher hand-over threshold
SCENES="bust atrium" FRAMES=0 timeout 900 bash tests/tools/ab_variants.sh "product below19 below20 fb2" 2 > gpurun_out/r06d/finish_below_ab.log 2>&1; cat gpurun_out/r06d/finish_below_ab.log
timeout 600 python tests/tools/shard_rate.py > gpurun_out/r06d/shard_rate.log 2>&1; cat gpurun_out/r06d/shard_rate.log
