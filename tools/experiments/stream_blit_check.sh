# are the streamed uploads shader (blit) copies or SDMA copies?  kernel statistics of the slow and of a fast layout (strict timeouts: rocprofv3
# with --memory-copy-trace hung on this path once)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "4 3 2" "4 0 2"; do set -- $cfg
rm -rf $R/gpurun_out/prof_blit
EXTRA_STREAMS=$3 GPU_MAX_HW_QUEUES=$1 timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_blit -o b -- python3 $R/tools/stream_timeline.py 12 lane_priority=$2 --no-marks 2>/dev/null | grep "clips/s" | sed "s/^/queues $1 lane_priority $2 other streams $3: /"
f=$(find $R/gpurun_out/prof_blit -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && (head -1 $f | cut -c1-80; grep -i "copy\|fill\|rocclr" $f | cut -c1-160; sed -n 2,4p $f | cut -c1-120)
done
