"""conv1 tuning aid: time the three conv1 launches on the production geometry (8 clips x 150 frames, 154 positions, masked or
dense frames) under the JG_CONV1_DBG / JG_CONV1_PRIO environment switches.  Usage: [env] python tools/conv1_ablate.py [dense]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.gestsync import GestSync
eng = Engine(0)
GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
B = int(os.environ.get("JG_ABL_CLIPS", "32"))
dense = len(sys.argv) > 1 and sys.argv[1] == "dense"
frames = torch.randint(0, 256, (B, 150, 270, 480, 3), dtype=torch.uint8, device="cuda") if dense else torch.from_numpy(synth.synth_frames(1234, B, 150)).cuda()
for _ in range(2):
    eng.debug_conv1_pool(frames, 4)
torch.cuda.synchronize()
eng.profile_reset(); eng.profile(True)
n = 5
for _ in range(n):
    eng.debug_conv1_pool(frames, 4)
p = eng.profile_get()
print("%s DBG=%s : conv1 kernel %.3f ms, scan+edge %.3f ms per %d clips  (x4 = %.2f ms per 32)" % (
    "dense" if dense else "masked", os.environ.get("JG_CONV1_DBG", "0"), 
    p["conv1"][0] / n, p["conv1_aux"][0] / n, B, 4 * p["conv1"][0] / n))
