"""Per-tile phase timeline of the persistent LDS-DMA GEMM (debug option gemm_timeline): prints, for workgroup 0
and the last workgroup, how long each tile spent in the k loop, issuing the next tile's first DMA, issuing
the epilogue and waiting for the next tile's data."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jegal_amd._lib import Engine
eng = Engine(0)
for o in sys.argv[1:]:
    k, v = o.split('='); eng.set_option(k, int(v))
M = int(os.environ.get("GEMM_M", 100800))
for (N, K, mode, name) in [(1536, 512, 0, "qkv"), (512, 512, 2, "out_proj+res"), (512, 512, 10, "out_proj+res+LN"), (512, 2048, 10, "linear2+res+LN")]:
    eng.debug_gemm(M, N, K, mode, 2)       # warm
    eng.set_option("gemm_timeline", 1)
    sys.stderr.write(f"==== {name} N={N} K={K}\n"); sys.stderr.flush()
    eng.debug_gemm(M, N, K, mode, 1)
    eng.set_option("gemm_timeline", 0)
