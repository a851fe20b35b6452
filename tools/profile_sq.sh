#!/bin/bash
# SQ / GRBM counter passes of ONE one-stream step of the bench command (run on the GPU box through gpurun):
#   tools/profile_sq.sh r4
# Three separate runs (SQ has 8 slots, GRBM 2; --kernel-trace only beside --pmc, the program directly behind `--`):
#   mfma : SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
#   lds  : SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
#   issue: SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
# tools/sq_summary.py condenses them into profiles/<tag>_sq_summary.json (per kernel: MFMA-busy fraction, implied clock,
# LDS-conflict fraction, issue-stall split).
set -u
TAG=${1:-r4}
shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
EXTRA=(--no-cpu-baseline --no-extras --steps 1 --warmup 1 --opt dual_stream=0 "$@")
run() { local name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- python3 "$ROOT/bench.py" "${EXTRA[@]}" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name rc=$?"; }
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
run issue SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
cd "$ROOT"
f() { find "$OUT/$1" -name "*$2" | head -1; }
python3 tools/sq_summary.py "$TAG" "$(f mfma counter_collection.csv)" "$(f lds counter_collection.csv)" "$(f issue counter_collection.csv)"
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_sq_summary.json gpurun_out/profiles_$TAG/ 2>/dev/null
tail -3 "$OUT"/*.err
