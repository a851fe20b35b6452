"""Is the per-stage HIP-event timing (jg_profile_*) consistent with the wall clock?  conv1 stage alone and inside the full step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.gestsync import GestSync
from jegal_amd.jegal import JEGAL
eng = Engine(0)
GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
JEGAL(engine=eng).load_state_dict(synth.jegal_state_dict())
frames = torch.from_numpy(synth.synth_frames(1234, 32, 150)).cuda()
out = torch.empty(32, 150, 512, device="cuda")
n = 10
for _ in range(3): eng.debug_conv1_pool(frames, 4)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): eng.debug_conv1_pool(frames, 4)
torch.cuda.synchronize()
print("conv1 alone: wall per call (scan+mask+conv1+edge) %.3f ms" % ((time.perf_counter() - t0) / n * 1e3))
eng.profile_reset(); eng.profile(True)
for _ in range(n): eng.debug_conv1_pool(frames, 4)
p = eng.profile_get(); eng.profile(False)
print("conv1 alone: events conv1 %.3f ms, aux %.3f ms" % (p["conv1"][0] / n, p["conv1_aux"][0] / n))
for _ in range(3): eng.extract_gesture(frames, out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): eng.extract_gesture(frames, out)
torch.cuda.synchronize()
print("full step: wall %.3f ms" % ((time.perf_counter() - t0) / n * 1e3))
eng.profile_reset(); eng.profile(True)
t0 = time.perf_counter()
for _ in range(n): eng.extract_gesture(frames, out)
torch.cuda.synchronize()
w = (time.perf_counter() - t0) / n * 1e3
p = eng.profile_get(); eng.profile(False)
print("full step with events: wall %.3f ms, sum of stages %.3f ms" % (w, sum(v[0] for v in p.values()) / n))
print({k: round(v[0] / n, 3) for k, v in p.items()})
