#!/bin/bash
# All rocprofv3 evidence of a round in one go (run on the GPU box through gpurun; outputs under gpurun_out/prof_<tag>/):
#   tools/profile_round.sh r4      (tools/profile_sq.sh <tag> adds the SQ / GRBM counter passes)
# 1. kernel trace + stats of the default bench command (two lanes) and of the one-stream mode,
# 2. two counter passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of one one-stream step,
# 3. kernel trace + stats of the config-3 (B = 64 vta) measurement and of the XLM-R front end,
# then tools/pmc_summary.py condenses 1 and 2 into profiles/<tag>_kernel_summary.csv / <tag>_pmc_summary.json.
set -u
TAG=${1:-r4}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift; rocprofv3 "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 "$ROOT/bench.py" --no-cpu-baseline "${EXTRA[@]}" > "$OUT/$name.json" 2> "$OUT/$name.err"; }
EXTRA=(--no-extras);                                  run dual --kernel-trace --stats
EXTRA=(--no-extras --opt dual_stream=0);              run single --kernel-trace --stats
EXTRA=(--no-extras --steps 1 --warmup 1 --opt dual_stream=0); run fetch --kernel-trace --pmc FETCH_SIZE
EXTRA=(--no-extras --steps 1 --warmup 1 --opt dual_stream=0); run write --kernel-trace --pmc WRITE_SIZE
EXTRA=(--only config3 --steps 5);                     run vta --kernel-trace --stats
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/xlmr" -o xlmr -- python3 "$ROOT/tools/xlmr_bench.py" 256 64 > "$OUT/xlmr.txt" 2> "$OUT/xlmr.err"
cd "$ROOT"
f() { find "$OUT/$1" -name "*$2" | head -1; }
python3 tools/pmc_summary.py "$TAG" "$(f single kernel_trace.csv)" "$(f fetch counter_collection.csv)" "$(f write counter_collection.csv)"
python3 tools/pmc_summary.py "${TAG}_vta" "$(f vta kernel_trace.csv)"
for n in dual single vta xlmr; do cp "$(f $n kernel_stats.csv)" "$OUT/${n}_kernel_stats.csv" 2>/dev/null; done
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_kernel_summary.csv profiles/${TAG}_pmc_summary.json profiles/${TAG}_vta_kernel_summary.csv gpurun_out/profiles_$TAG/ 2>/dev/null
cp "$OUT"/*_kernel_stats.csv "$OUT"/*.json "$OUT"/xlmr.txt gpurun_out/profiles_$TAG/ 2>/dev/null
ls -la gpurun_out/profiles_$TAG
