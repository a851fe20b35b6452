import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.jegal import JEGAL
eng = Engine(0)
JEGAL(engine=eng).load_state_dict(synth.jegal_state_dict())
for o in sys.argv[1:]:
    k, v = o.split('=')
    eng.set_option(k, int(v))
x = torch.randn(672, 150, 1024, device="cuda")          # 100800 tokens like the GestSync transformer
for _ in range(2):
    eng.jegal_gestures(x, None, align=True)
torch.cuda.synchronize()
eng.profile_reset(); eng.profile(True)
for _ in range(2):
    eng.jegal_gestures(x, None, align=True)
p = eng.profile_get()
print("%s JG_GEMM_DBG=%s gemm ms per call: %.3f (%d launches)" % (" ".join(sys.argv[1:]), os.environ.get("JG_GEMM_DBG", "0"), p["gemm"][0] / 2, p["gemm"][1] / 2))
