import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jegal_amd._lib import Engine
eng = Engine(0)
for o in sys.argv[1:]:
    k, v = o.split('='); eng.set_option(k, int(v))
for (M, N, K, mode, name) in [(3646720, 128, 1600, 4, "conv2-shape plain"), (936320, 256, 1152, 4, "conv3-shape plain"), (492800, 256, 2304, 4, "conv4/5-shape plain"),
                              (4928, 512, 4096, 4, "fc6")]:
    ms = eng.debug_gemm(M, N, K, mode, 5)
    print(f"{name:22s} M={M:8d} N={N:4d} K={K:5d}  {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s")
