"""Operand-precision emulation of the HIP pipeline across WEIGHT FAMILIES, on CPU (test infrastructure; build container).

VERDICT r4 items 1-2: which precision treatment of the Linear weights holds 1e-3 on weights that are not one Gaussian draw, and
is a correction term built from the clip's OWN rows (no calibration pass) as good as hi+lo weights?

Every conv / Linear / matmul operand is rounded to fp16, accumulation fp32, LayerNorm / softmax / residual fp32 (DESIGN.md
section 3).  Linear weights (the conv weights stay single fp16 in every mode, as in the product):
  fp16   single fp16 weights
  w2     hi + lo fp16 pair (the product's JG_PREC_FP16_W2)
  bc     single fp16 + (w - fp16(w)) . E[x] folded into the bias, E[x] recorded on a DIFFERENT clip (the product's JG_PREC_FP16_BC
         with its built-in calibration clips)
  rc     single fp16 + (w - fp16(w)) . mean of THIS clip's rows of x, per Linear call (JG_PREC_FP16_RC, round 5)
Prints rel-L2 of the GestSync features and the unit-norm gesture embedding vs the fp32 oracle.

    python oracle/precision_families.py [T] [family ...]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import jegal_oracle as O
from jegal_amd import synth


def q16(x):
    return x.to(torch.float16).float()


class Emu:
    def __init__(self, mode):
        self.mode = mode
        self.record = None          # list of E[x] per Linear call (calibration pass) or None
        self.replay = None
        self.idx = 0
        self.maxabs = 0.0

    def linear(self, lin):
        def f(x, w, b=None):
            self.maxabs = max(self.maxabs, float(x.abs().max()))
            xq = q16(x)
            wh = q16(w)
            if self.mode == "fp16":
                return lin(xq, wh, b)
            lo = q16(w - wh)
            if self.mode == "w2":
                return lin(xq, wh, b) + lin(xq, lo)
            mu = xq.reshape(-1, xq.shape[-1]).mean(0)
            if self.record is not None:
                self.record.append(mu)
                return lin(xq, wh, b) + lin(xq, lo)
            if self.mode == "bc":
                mu = self.replay[self.idx]
                self.idx += 1
            corr = lo @ mu
            return lin(xq, wh, b) + corr
        return f


def run(family, seed_off, T, modes):
    gsd = O.tensors(synth.gestsync_state_dict(seed=synth.GESTSYNC_SEED + seed_off, include_unused=False, family=family))
    jsd = O.tensors(synth.jegal_state_dict(seed=synth.JEGAL_SEED + seed_off, family=family))
    frames = synth.synth_frames(1234, 1, T)[0]
    calib = synth.synth_frames(99, 1, T)[0]
    to01 = lambda f: torch.from_numpy(f.astype(np.float32) / np.float32(255.0))

    def pipeline(f01):
        feats = O.gestsync_clip_feats(gsd, f01, naive=False)
        g = O.jegal_forward_inference(jsd, visual_feats=feats[None], visual_mask=torch.ones(1, T))
        return feats, O.l2_normalize(g[0])

    rel = lambda a, b: float((a - b).norm() / b.norm())
    with torch.no_grad():
        ref_f, ref_g = pipeline(to01(frames))
        lin, c3, mm = F.linear, F.conv3d, torch.matmul
        out = {}
        for mode in modes:
            emu = Emu(mode)
            F.conv3d = lambda x, w, b=None, **k: c3(q16(x), q16(w), b, **k)
            torch.matmul = lambda a, b: mm(q16(a), q16(b))
            F.linear = emu.linear(lin)
            try:
                if mode == "bc":
                    emu.record = []
                    pipeline(to01(calib))
                    emu.replay, emu.record = emu.record, None
                f, g = pipeline(to01(frames))
            finally:
                F.linear, F.conv3d, torch.matmul = lin, c3, mm
            out[mode] = (rel(f, ref_f), rel(g, ref_g), float((g - ref_g).abs().max()), emu.maxabs)
    return out


if __name__ == "__main__":
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    fams = sys.argv[2:] or ["gauss", "gauss+1", "gauss+2", "heavy", "sharp"]
    modes = ["fp16", "w2", "bc", "rc"]
    print(f"T = {T}; columns: GestSync feats rel-L2 | gesture emb rel-L2 | emb max-abs | max |x| into a Linear")
    for fam in fams:
        name, _, off = fam.partition("+")
        res = run(name, int(off or 0), T, modes)
        for m in modes:
            a, b, c, d = res[m]
            print(f"{fam:9s} {m:5s} feats {a:.2e} | emb {b:.2e} | max-abs {c:.2e} | max|x| {d:.1f}")
