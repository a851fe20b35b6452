import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jegal_amd._lib import Engine
eng = Engine(0)
for o in sys.argv[1:]:
    k, v = o.split('='); eng.set_option(k, int(v))
M = int(os.environ.get("GEMM_M", 100800))
print("options:", sys.argv[1:])
for (N, K, mode, name) in [(1536, 512, 0, "qkv"), (512, 512, 2, "out_proj+res"), (2048, 512, 4, "linear1+relu"), (512, 2048, 2, "linear2+res"), (512, 512, 10, "out_proj+res+LN"), (512, 2048, 10, "linear2+res+LN"),
                           (512, 512, 0, "512x512 f16out"), (512, 1024, 0, "K=1024"), (512, 4096, 0, "K=4096"), (2048, 2048, 0, "2048x2048")]:
    ms = eng.debug_gemm(M, N, K, mode, 10)
    fl = 2.0 * M * N * K
    print(f"{name:18s} N={N:5d} K={K:5d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s")
