"""Lane split of the 32-clip batch (option dual_split32: the first lane gets this many clips of 32) against the step time, one box,
each value a fresh bench.py process:  python tools/dual_split_sweep.py [bench.py options ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(2):
    row = []
    for v in (12, 14, 15, 16, 17, 18, 20):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--steps", "100", "--opt", f"dual_split32={v}"] + sys.argv[1:],
                             capture_output=True, text=True).stdout.strip().splitlines()
        row.append(f"{v}:{json.loads(out[-1])['ms_per_step']:.3f}")
    print("ms per step by clips in the first lane:", " ".join(row), flush=True)
