"""Where does the fp16 modes' error come from?  

Runs full-length clips (T = 150) through engines that kept the fp32 matrices (option audit_weights) and moves ONE stage at a time to the
fp32 audit kernels (option audit_stages: 1 conv stack, 2 GestSync transformer + ff_vid, 4 JEGAL gesture branch), next to the plain
precision modes, everything against the CPU oracle.  Prints the table and writes gpurun_out/r6_precision_floor.json.

    python tools/precision_floor.py [--families gauss+0,gauss+1,heavy+0] [--clips 2]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import jegal_oracle as O  # noqa: E402
from jegal_amd import synth  # noqa: E402
import jegal_amd._lib as L  # noqa: E402
from jegal_amd.gestsync import GestSync  # noqa: E402
from jegal_amd.jegal import JEGAL  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--families", default="gauss+0,gauss+1,heavy+0")
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--frames", type=int, default=150)
    ap.add_argument("--opts", default="", help="extra engine options for the RC rows, e.g. stream_fp16=0")
    ap.add_argument("--rc-ablate", action="store_true", help="rows with the run-time correction switched off per Linear type")
    ap.add_argument("--quick", action="store_true", help="only the four rows of round 6's measures")
    args = ap.parse_args()
    T, B = args.frames, args.clips
    frames = synth.synth_frames(1234, B, T)
    dev = torch.from_numpy(frames).cuda()
    table = {}
    for fam in args.families.split(","):
        name, off = fam.split("+")
        off = int(off)
        gsd = synth.gestsync_state_dict(seed=synth.GESTSYNC_SEED + off, include_unused=False, family=name)
        jsd = synth.jegal_state_dict(seed=synth.JEGAL_SEED + off, family=name)
        gt, jt = O.tensors(gsd), O.tensors(jsd)
        refs = []
        with torch.no_grad():
            for b in range(B):
                f = O.gestsync_clip_feats(gt, torch.from_numpy(frames[b].astype(np.float32) / np.float32(255.0)))
                g = O.l2_normalize(O.jegal_forward_inference(jt, visual_feats=f[None], visual_mask=torch.ones(1, T))[0])
                refs.append((f.numpy(), g.numpy()))
        rows = {}
        extra = [tuple([kv.split("=")[0], int(kv.split("=")[1])]) for kv in args.opts.split(",") if kv]

        def run(tag, prec, aw=False, masks=(0,), opts=(), parts=(0,)):
            e = L.Engine(0, precision=prec)
            try:
                if aw:
                    e.set_option("audit_weights", 1)
                for k, v in opts:
                    e.set_option(k, v)
                GestSync(engine=e).load_state_dict(gsd)
                JEGAL(engine=e).load_state_dict(jsd)
                for m, jp in [(m, jp) for m in masks for jp in parts]:
                    if aw:
                        e.set_option("audit_stages", m)
                        e.set_option("audit_jegal_parts", jp)
                    emb = e.extract_gesture(dev).cpu().numpy()
                    feats = e.gestsync_clip(dev).cpu().numpy()
                    key = tag if not aw else f"{tag} stages={m}" + (f" jegal_parts={jp}" if jp else "")
                    rows[key] = {"gesture_rel": max(rel(emb[b], refs[b][1]) for b in range(B)),
                                 "gesture_maxabs": max(float(np.abs(emb[b] - refs[b][1]).max()) for b in range(B)),
                                 "feats_rel": max(rel(feats[b], refs[b][0]) for b in range(B))}
                    print(f"[{fam}] {key:32s} gesture rel-L2 {rows[key]['gesture_rel']:.3e} max-abs {rows[key]['gesture_maxabs']:.3e} | "
                          f"GestSync feats {rows[key]['feats_rel']:.3e}", flush=True)
            finally:
                e.close()

        # round 6's two measures, alone and together (the default): the JEGAL branch's ends on the fp32 kernel, conv weights rounded with
        # per-channel error diffusion across the taps
        run("rc r5 (ends=0 diffuse=0)", L.PREC_FP16_RC, opts=[("jegal_fp32_ends", 0), ("conv_round_diffuse", 0)])
        run("rc ends=1 diffuse=0", L.PREC_FP16_RC, opts=[("conv_round_diffuse", 0)])
        run("rc ends=0 diffuse=1", L.PREC_FP16_RC, opts=[("jegal_fp32_ends", 0)])
        run("rc default", L.PREC_FP16_RC)
        run("rc default + jegal_ffn_x3", L.PREC_FP16_RC, opts=[("jegal_ffn_x3", 1)])
        if args.rc_ablate:
            # which Linear types of the GestSync transformer need the run-time correction?  (option rc_layers: 1 qkv, 2 out_proj, 4 linear1 +
            # ff_vid.0, 8 linear2; the others run single fp16 without a correction -- measurement only)
            for m in (0, 1, 2, 4, 8, 14, 13, 11, 7, 12, 10, 6):
                run(f"rc rc_layers={m}", L.PREC_FP16_RC, opts=[("rc_layers", m)])
        if args.quick:
            table[fam] = rows
            continue
        # what remains in the round-6 default, by stage
        run("rc default", L.PREC_FP16_RC, aw=True, masks=(1, 2, 4, 3, 5, 6, 7))
        extra = extra + [("jegal_fp32_ends", 0), ("conv_round_diffuse", 0)]          # the decomposition below is of the round-5 arithmetic
        run("rc", L.PREC_FP16_RC, aw=True, masks=(0, 1, 2, 4, 3, 5, 6, 7), opts=extra)
        # inside the JEGAL branch, on exact GestSync features (stages 3): 1 input projection, 2 attention sub-layers, 4 feed-forward
        # sub-layers, 8 final norm + output / align projections in fp32
        run("rc", L.PREC_FP16_RC, aw=True, masks=(3,), parts=(1, 2, 4, 8, 6, 14, 15), opts=extra)
        run("w2", L.PREC_FP16_W2, aw=True, masks=(0, 1, 2, 4), opts=extra)
        run("w2_all", L.PREC_FP16_W2_ALL, opts=extra)
        run("fp32", L.PREC_FP32)
        table[fam] = rows
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r6_precision_quick.json" if args.quick else "r6_precision_floor.json"), "w") as f:      # (--quick used to overwrite the full table)
        json.dump(table, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
