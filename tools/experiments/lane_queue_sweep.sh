for n in 0 1 2 3 5 8; do for lp in 0 3; do
python bench.py --no-extras --no-cpu-baseline --steps 60 --extra-streams $n --opt lane_priority=$lp > gpurun_out/lp_${n}_${lp}.json 2>gpurun_out/lp_${n}_${lp}.err
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/lp_${n}_${lp}.json").read().strip().splitlines()[-1]); print("extra streams ${n} lane_priority ${lp}:", d["value"], "clips/s", d["ms_per_step"], "ms")
except Exception as e:
    print("extra streams ${n} lane_priority ${lp}: FAILED", e, open("gpurun_out/lp_${n}_${lp}.err").read()[-300:])
PY
done; done
