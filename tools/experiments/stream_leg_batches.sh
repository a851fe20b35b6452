for cfg in "3 8" "3 4" "0 8"; do set -- $cfg
GPU_MAX_HW_QUEUES=$2 python bench.py --no-cpu-baseline --opt lane_priority=$1 2>&1 >/dev/null | grep "source-resolution stream" | sed "s/^/lane_priority $1 queues $2: /"
done
