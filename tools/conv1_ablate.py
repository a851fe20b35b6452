import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.gestsync import GestSync
eng = Engine(0)
GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
frames = torch.from_numpy(synth.synth_frames(1234, 4, 150)).cuda()
for _ in range(2):
    eng.debug_conv1_pool(frames, 12)
torch.cuda.synchronize()
eng.profile_reset(); eng.profile(True)
for _ in range(3):
    eng.debug_conv1_pool(frames, 12)
p = eng.profile_get()
print("JG_CONV1_DBG=%s conv1 ms per 4 clips: %.3f  (pool %.3f)" % (os.environ.get("JG_CONV1_DBG", "0"), p["conv1"][0] / p["conv1"][1], p["maxpool"][0] / max(1, p["maxpool"][1])))
