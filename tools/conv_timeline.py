"""Per-tile phase timeline (option gemm_timeline) of every GEMM/conv launch of one 32-clip step; filter stderr."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.gestsync import GestSync
eng = Engine(0); eng.set_chunk(32)
GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
frames = torch.from_numpy(synth.synth_frames(1234, 32, 150)).cuda()
eng.gestsync_clip(frames)
torch.cuda.synchronize()
eng.set_option("gemm_timeline", 1)
eng.gestsync_clip(frames)
eng.set_option("gemm_timeline", 0)
