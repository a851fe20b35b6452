"""A/B two builds of libjegal_hip.so on the XLM-R front end (tools/xlmr_bench.py), alternating on ONE box (boxes differ by +-4 %).
Usage (on the GPU box):  python tools/ab_xlmr.py <old.so> [pairs=3] [xlmr_bench.py arguments ...]     (default: 256 64)"""
import os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "jegal_amd", "libjegal_hip.so")


def run(tag, extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "xlmr_bench.py")] + extra + ["--no-cpu"], capture_output=True, text=True).stdout
    ms = [float(x) for x in re.findall(r"([0-9.]+) ms per batch", out)]
    print(tag, " ".join("%.3f" % v for v in ms), "ms per batch (hi+lo, calibrated)", flush=True)
    return ms


def main():
    old = sys.argv[1]
    pairs = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
    extra = sys.argv[3:] if len(sys.argv) > 2 and sys.argv[2].isdigit() else sys.argv[2:]
    extra = extra or ["256", "64"]
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
    for i, name in enumerate(("hi+lo", "calibrated")):
        mo, mn = sum(r[i] for r in res["old"]) / pairs, sum(r[i] for r in res["new"]) / pairs
        print("%s: mean old %.3f, new %.3f ms per batch (%+.2f %%)" % (name, mo, mn, 100 * (mn / mo - 1)))


if __name__ == "__main__":
    main()
