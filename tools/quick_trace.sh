#!/bin/bash
# One-stream kernel trace of a few bench steps -> profiles/<tag>_kernel_summary.csv (tools/pmc_summary.py).  tools/quick_trace.sh <tag> [bench args]
set -u
TAG=${1:-q}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/single" -o single -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras --steps 3 --warmup 2 --opt dual_stream=0 "$@" > "$OUT/single.json" 2> "$OUT/single.err"
cd "$ROOT"
python3 tools/pmc_summary.py "$TAG" "$(find "$OUT/single" -name '*kernel_trace.csv' | head -1)"
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_kernel_summary.csv gpurun_out/profiles_$TAG/
cat profiles/${TAG}_kernel_summary.csv
