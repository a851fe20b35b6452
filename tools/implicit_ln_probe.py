"""Would the GestSync transformer gain from XLM-R's implicit LayerNorm?  The four Linear shapes of a layer at M = 100 800 tokens on
random operands: today's kernels (plain 256x256 with fp16 out; LN-fused 128x512) against the implicit-LayerNorm consumer / producer
instances of the 256x256 tile (timing only).  Usage: python tools/implicit_ln_probe.py [M]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jegal_amd._lib import Engine
eng = Engine(0)
torch.manual_seed(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 100800
rows = [("qkv", 512, 1536, [("plain fp16 out", 0), ("implicit-LN consumer", 16)]),
        ("out_proj", 512, 512, [("LN-fused 128x512", 8), ("plain fp16 out", 0), ("implicit-LN producer", 32)]),
        ("linear1 (ReLU)", 512, 2048, [("plain fp16 out", 4), ("implicit-LN consumer", 20)]),
        ("linear2", 2048, 512, [("LN-fused 128x512", 8), ("plain fp16 out", 0), ("implicit-LN producer", 32)])]
for name, K, N, variants in rows:
    a = (torch.randn((M, K), device="cuda") * 0.5).half()
    w = (torch.randn((N, K), device="cuda") * 0.05).half()
    out = []
    for tag, mode in variants:
        ts = sorted(eng.debug_gemm(M, N, K, mode=mode, iters=20, a16=a, w16=w) for _ in range(3))
        out.append(f"{tag} {ts[1] * 1e3:6.1f} us")
    print(f"{name:15s} M={M} K={K:4d} N={N:4d}: " + " | ".join(out), flush=True)
