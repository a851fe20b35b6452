"""Per-tile-shape timing of the plain LDS-DMA GEMM on the small-M shapes (JEGAL branch, XLM-R): calibrates the cost estimate of
launch_glds (gemm.hip).  Usage: python tools/gemm_tiles.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jegal_amd._lib import Engine
eng = Engine(0)
torch.manual_seed(0)
SH = [(4800, 512, 512, 0), (4800, 512, 1536, 0), (4800, 512, 2048, 4), (4800, 2048, 512, 2), (4800, 1024, 512, 2), (9600, 512, 2048, 4), (9600, 2048, 512, 2),
      (16384, 768, 2304, 0), (16384, 768, 768, 2), (16384, 768, 3072, 4), (16384, 3072, 768, 2), (2048, 768, 2304, 0), (2048, 3072, 768, 2),
      (768, 768, 3072, 1), (768, 3072, 768, 3), (640, 512, 512, 1)]
for M, K, N, mode in SH:
    a = (torch.randn((M, K), device="cuda") * 0.5).half()
    w = (torch.randn((N, K), device="cuda") * 0.05).half()
    res = []
    for tile in (0, 1, 2, 3):
        eng.set_option("gemm_tile", tile)
        ts = sorted(eng.debug_gemm(M, N, K, mode=mode, iters=20, a16=a, w16=w) for _ in range(3))
        res.append(ts[1] * 1e3)
    eng.set_option("gemm_tile", 0)
    gf = 2.0 * M * N * K / 1e9
    print(f"M={M:6d} K={K:5d} N={N:5d} mode={mode}: auto {res[0]:7.1f} us ({gf / res[0] * 1e3:5.0f} TF) | 128x128 {res[1]:7.1f} | 256x128 {res[2]:7.1f} | 256x256 {res[3]:7.1f}", flush=True)
