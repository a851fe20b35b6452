"""XLM-RoBERTa GEMM shapes (M = B*L tokens, d 768, ffn 3072) one by one: time per tile shape on random operands and the per-tile
phase timeline of the automatic choice.  Usage: python tools/xlmr_gemm_probe.py [M]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jegal_amd._lib import Engine
eng = Engine(0)
torch.manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
SH = [("qkv", 768, 2304, 0), ("out+res", 768, 768, 2), ("ff1", 768, 3072, 0), ("ff2+res", 3072, 768, 2)]
for name, K, N, mode in SH:
    a = (torch.randn((M, K), device="cuda") * 0.5).half()
    w = (torch.randn((N, K), device="cuda") * 0.05).half()
    res = []
    for tile in (0, 1, 2, 3):
        eng.set_option("gemm_tile", tile)
        ts = sorted(eng.debug_gemm(M, N, K, mode=mode, iters=20, a16=a, w16=w) for _ in range(3))
        res.append(ts[1] * 1e3)
    eng.set_option("gemm_tile", 0)
    gf = 2.0 * M * N * K / 1e9
    print(f"{name:8s} M={M:6d} K={K:5d} N={N:5d} mode={mode}: auto {res[0]:7.1f} us ({gf / res[0] * 1e3:5.0f} TF) | 128x128 {res[1]:7.1f} | 256x128 {res[2]:7.1f} | 256x256 {res[3]:7.1f}", flush=True)
    eng.set_option("gemm_timeline", 1)
    sys.stderr.write(f"==== {name} N={N} K={K}\n"); sys.stderr.flush()
    eng.debug_gemm(M, N, K, mode=mode, iters=1, a16=a, w16=w)
    eng.set_option("gemm_timeline", 0)
