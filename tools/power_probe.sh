#!/bin/bash
# Sample board power and shader clock (rocm-smi, read-only) while the bench's timed loop runs: is the path power-bound?
#   tools/power_probe.sh [bench.py options]   -> gpurun_out/power_probe.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/power_probe.txt
mkdir -p "$ROOT/gpurun_out"
python3 "$ROOT/bench.py" --no-cpu-baseline --no-extras --steps 600 "$@" > "$ROOT/gpurun_out/power_probe_bench.json" 2>/dev/null &
BP=$!
: > "$OUT"
while kill -0 $BP 2>/dev/null; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks --json 2>/dev/null | tr -d '\n' >> "$OUT"
    echo >> "$OUT"
    sleep 0.25
done
wait $BP
python3 - "$OUT" <<'PY'
import json, sys
rows = []
for ln in open(sys.argv[1]):
    ln = ln.strip()
    if not ln.startswith("{"):
        continue
    try:
        d = json.loads(ln)
    except Exception:
        continue
    c = d.get("card0", {})
    p = [v for k, v in c.items() if "ower" in k and "W" in k]
    s = [v for k, v in c.items() if "sclk" in k]
    rows.append((p[0] if p else None, s[0] if s else None))
print("samples:", len(rows))
for r in rows:
    print(r)
PY
