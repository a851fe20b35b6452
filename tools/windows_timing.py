"""Time the literal forward_vid drop-in (jg_gestsync_windows): N = 48 fp32 windows (N,3,25,270,480), the reference's batch
(inference_embs.py:499).  This path materialises what the clip path never does (fp32 windows -> stacked fp16 frames -> implicit-GEMM
conv1 -> separate max-pool): it exists for callers that hold windows, not for throughput.  Usage: python tools/windows_timing.py [N]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.gestsync import GestSync

N = int(sys.argv[1]) if len(sys.argv) > 1 else 48
eng = Engine(0)
gs = GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
x = torch.rand((N, 3, 25, 270, 480), device="cuda")
x[:, :, :, :110] = 0
for _ in range(2):
    gs.forward_vid(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for _ in range(n):
    out = gs.forward_vid(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("forward_vid: %d windows in %.2f ms = %.0f windows/s (= %.1f clips/s of 150 frames if every window were evaluated this way); out %s"
      % (N, dt * 1e3, N / dt, N / dt / 150, tuple(out.shape)))
