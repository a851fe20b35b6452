"""A/B an engine option on ONE box, alternating runs of bench.py (boxes differ by +-3 %, so only same-call pairs are credible).
  python tools/ab_option.py <name> <valueA> <valueB> [pairs=3] [bench.py options ...]     e.g.  ab_option.py stream_fp16 0 1 3 --precision 5"""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(tag, extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", "--steps", "100"] + extra,
                         capture_output=True, text=True).stdout.strip().splitlines()
    d = json.loads(out[-1])
    s = d["stage_ms_per_step"]
    print(tag, "%.3f ms/step" % d["ms_per_step"], " ".join("%s %.3f" % (k, v) for k, v in s.items() if v), flush=True)
    return d["ms_per_step"]


def main():
    name, va, vb = sys.argv[1:4]
    pairs = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4].isdigit() else 3
    extra = sys.argv[5:] if len(sys.argv) > 4 and sys.argv[4].isdigit() else sys.argv[4:]
    res = {va: [], vb: []}
    for _ in range(pairs):
        for v in (va, vb):
            res[v].append(run(f"{name}={v}", extra + ["--opt", f"{name}={v}"]))
    ma, mb = sum(res[va]) / pairs, sum(res[vb]) / pairs
    print("mean %s=%s %.3f, %s=%s %.3f ms/step (%+.2f %%)" % (name, va, ma, name, vb, mb, 100 * (mb / ma - 1)))


if __name__ == "__main__":
    main()
