"""A/B two builds of libjegal_hip.so on ONE box: boxes differ by +-3 % (12.0-12.6 ms per step), more than most changes are worth,
so a change is only credible when the old and the new library alternate in the same gpurun call.

Usage (on the GPU box):  python tools/ab_bench.py <old.so> [pairs=3] [bench.py options ...]
  old.so: the previous build, e.g. copied to tools/bin/ before rebuilding (tools/bin/ is git-ignored but travels with gpurun);
  the current jegal_amd/libjegal_hip.so is "new".  Prints ms per step and the one-stream stage table of every run."""
import json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "jegal_amd", "libjegal_hip.so")


def run(tag, extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--steps", "100"] + extra,
                         capture_output=True, text=True).stdout.strip().splitlines()
    d = json.loads(out[-1])
    s = d["stage_ms_per_step"]
    print(tag, "%.3f ms/step" % d["ms_per_step"], " ".join("%s %.3f" % (k, v) for k, v in s.items() if v), flush=True)
    return d["ms_per_step"]


def main():
    old = sys.argv[1]
    pairs = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
    extra = sys.argv[3:] if len(sys.argv) > 2 and sys.argv[2].isdigit() else sys.argv[2:]
    new = LIB + ".new"
    shutil.copy(LIB, new)
    res = {"old": [], "new": []}
    try:
        for _ in range(pairs):
            shutil.copy(old, LIB)
            res["old"].append(run("old", extra))
            shutil.copy(new, LIB)
            res["new"].append(run("new", extra))
    finally:
        shutil.copy(new, LIB)
        os.remove(new)
    mo, mn = sum(res["old"]) / pairs, sum(res["new"]) / pairs
    print("mean old %.3f, new %.3f ms/step (%+.2f %%)" % (mo, mn, 100 * (mn / mo - 1)))


if __name__ == "__main__":
    main()
