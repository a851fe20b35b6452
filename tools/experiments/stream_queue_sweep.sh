# streamed (PCIe-inclusive, decoder-resolution) throughput against GPU_MAX_HW_QUEUES, the lanes' priority and the number of other streams in the process
for q in 4 8; do for lp in 0 3; do for x in 0 2 4 6; do
echo -n "GPU_MAX_HW_QUEUES=$q lane_priority=$lp other streams $x: "
EXTRA_STREAMS=$x GPU_MAX_HW_QUEUES=$q timeout 120 python tools/stream_timeline.py 12 lane_priority=$lp --no-marks 2>&1 | grep "clips/s"
done; done; done
