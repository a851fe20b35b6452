"""LN-fused GEMM: launch time by JD (the epilogue row block in front of which the next tile's first DMA goes; -DJG_LNF_JD builds in tools/bin/)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import sys, torch
sys.path.insert(0, %r)
import jegal_amd._lib as L
L.LIB_PATH = sys.argv[1]
from jegal_amd._lib import Engine
eng = Engine(0)
M = 100800
out = []
for K in (512, 2048):
    a = (torch.rand((M, K), device="cuda") - 0.5).half(); w = (torch.rand((512, K), device="cuda") - 0.5).half()
    eng.debug_gemm(M, 512, K, 8, 10, a, w)
    out.append(min(eng.debug_gemm(M, 512, K, 8, 20, a, w) for _ in range(3)) * 1e3)
print("out_proj %%.1f us  linear2 %%.1f us" %% tuple(out))
''' % ROOT
for rep in range(2):
    for name in ("jegal_amd/libjegal_hip.so (JD = 2)", "tools/bin/lib_jd1.so", "tools/bin/lib_jd3.so", "tools/bin/lib_jd4.so"):
        p = os.path.join(ROOT, name.split(" ")[0])
        r = subprocess.run([sys.executable, "-c", code, p], capture_output=True, text=True)
        print(f"{name:40s}", r.stdout.strip() or r.stderr.strip()[-200:], flush=True)
