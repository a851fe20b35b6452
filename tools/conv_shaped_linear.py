import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jegal_amd._lib import Engine
eng = Engine(0)
for (M, N, K, name) in [(492800, 256, 2304, "conv5-shaped linear"), (936320, 256, 1152, "conv3-shaped linear"), (3646720, 128, 1600 + 64 - 1600 % 64 if 1600 % 64 else 1600, "conv2-shaped linear")]:
    ms = eng.debug_gemm(M, N, K, 4, 5)
    print(f"{name:22s} M={M} N={N} K={K}: {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s")
