import sys, os
sys.path.insert(0, "/root/repo")
import torch
from jegal_amd._lib import Engine
eng = Engine(0)
M = 100800
for K in (512, 2048):
    a = (torch.rand((M, K), device="cuda") - 0.5).half(); w = (torch.rand((512, K), device="cuda") - 0.5).half()
    for rep in range(2):
        row = []
        for st in (-1, 0, 200, 350, 500, 700, 900, 1200, 1600, 2000):
            eng.set_option("gemm_stagger", st)
            ms = eng.debug_gemm(M, 512, K, 8, 20, a, w)
            row.append(f"{st}:{ms*1e3:.1f}")
        print(f"K={K} LN-fused us per launch by gemm_stagger (10-ns ticks per phase; 0 = auto):", " ".join(row), flush=True)
