import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import jegal_oracle as O
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.gestsync import GestSync
eng = Engine.get("cuda:0")
gs = GestSync(engine=eng).load_state_dict(synth.gestsync_state_dict())
g = np.load(os.path.join(ROOT, "tests/golden/gestsync_clip.npz"))
frames = synth.synth_frames(int(g["seed"]), 1, int(g["T"]))[0]
f01 = O.pad_clip(torch.from_numpy(frames.astype(np.float32) / np.float32(255.0)))
vol = f01.permute(3, 0, 1, 2)
x = torch.stack([vol[:, i:i + 25] for i in range(2)])
out, oc = gs.forward_vid(x.cuda(), return_feats=True)
out = out.cpu().numpy(); ref = g["out_full"]
print("shape", out.shape, "nan", np.isnan(out).sum(), "absmax", np.abs(out).max(), np.abs(ref).max())
d = np.abs(out - ref)
print("err by window", d.reshape(2, -1).max(1))
print("err by channel block", d.transpose(1, 0, 2).reshape(16, -1).max(1))
print("err by token", d.transpose(2, 0, 1).reshape(21, -1).max(1))
idx = np.unravel_index(np.argmax(d), d.shape); print("worst", idx, out[idx], ref[idx])
