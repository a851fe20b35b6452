"""Emulate the HIP pipeline's operand precision on CPU (test infrastructure; build container or GPU box host).

Every GEMM / conv operand (activation and weight) is rounded to the probe dtype, accumulation
stays fp32, the residual stream / LayerNorm / softmax stay fp32 -- the storage plan of DESIGN.md.
Prints rel-L2 error of the final unit-norm gesture embedding vs the fp32 oracle.
"""
import sys, os
import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import jegal_oracle as O
from jegal_amd import synth


def run(dtype, T=6):
    gsd = O.tensors(synth.gestsync_state_dict(include_unused=False))
    jsd = O.tensors(synth.jegal_state_dict())
    frames = synth.synth_frames(1234, 1, T)[0]
    f01 = torch.from_numpy(frames.astype(np.float32) / np.float32(255.0))

    def pipeline():
        feats = O.gestsync_clip_feats(gsd, f01, naive=False)
        g = O.jegal_forward_inference(jsd, visual_feats=feats[None], visual_mask=torch.ones(1, T))
        return feats, O.l2_normalize(g[0])

    with torch.no_grad():
        ref_f, ref_g = pipeline()
        if dtype is None:
            return
        q = lambda x: x.to(dtype).float()
        lin, c3, c2, mm = F.linear, F.conv3d, F.conv2d, torch.matmul
        F.linear = lambda x, w, b=None: lin(q(x), q(w), b)
        F.conv3d = lambda x, w, b=None, **k: c3(q(x), q(w), b, **k)
        torch.matmul = lambda a, b: mm(q(a), q(b))
        try:
            f, g = pipeline()
        finally:
            F.linear, F.conv3d, F.conv2d, torch.matmul = lin, c3, c2, mm
    rel = lambda a, b: float((a - b).norm() / b.norm())
    print(f"{dtype}: gestsync feats rel-L2 {rel(f, ref_f):.2e} | gesture emb rel-L2 {rel(g, ref_g):.2e} | max-abs {float((g-ref_g).abs().max()):.2e}")


if __name__ == "__main__":
    for dt in (torch.float16, torch.bfloat16):
        run(dt)
