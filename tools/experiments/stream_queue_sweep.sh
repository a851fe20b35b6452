# streamed (PCIe-inclusive, decoder-resolution) throughput against GPU_MAX_HW_QUEUES and the lanes' priority; three repeats each
for q in 4 8; do for lp in 0 3; do for r in 1 2 3; do
echo -n "GPU_MAX_HW_QUEUES=$q lane_priority=$lp: "
GPU_MAX_HW_QUEUES=$q python tools/stream_timeline.py 24 lane_priority=$lp --no-marks 2>&1 | grep "clips/s"
done; done; done
