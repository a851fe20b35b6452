"""Debug probe: run-to-run differences of the two-lane gesture path under a poisoned workspace (companion of xl_poison_probe.py)."""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.gestsync import GestSync
from jegal_amd.jegal import JEGAL
eng = Engine(0)
GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict(include_unused=False))
JEGAL(engine=eng).load_state_dict(synth.jegal_state_dict())
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (32, 30)
frames = torch.from_numpy(synth.synth_frames(5, B, T)).cuda()
base = eng.extract_gesture(frames).clone()
eng.set_option("ws_poison", 1)
n = 0
for it in range(150):
    eng.set_option("gemm_tile", it & 3)
    out = eng.extract_gesture(frames)
    if not torch.equal(out, base):
        n += 1
        d = (out - base).abs().amax(-1).cpu().numpy()
        print("it", it, "clips differing", sorted(set(int(b) for b, _ in np.argwhere(d > 0))), "max", float(d.max()))
print(f"gesture path B={B} T={T}, two lanes, poisoned workspace: runs with differences:", n, "of 150")
