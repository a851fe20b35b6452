"""Debug probe 2: what do the differing XLM-R outputs look like?  NaNs?  Equal to the no-mask result (mask buffer clobbered)?"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.xlmr import XLMRoberta
B, L = 64, 32
ids, mask = synth.xlmr_inputs(3, B, L)
lens = mask.sum(1)
eng = Engine(0)
eng.set_option("xlmr_lanes", 2)
xl = XLMRoberta(engine=eng).load_state_dict(synth.xlmr_state_dict(layers=int(os.environ.get("LAYERS", "2"))))
ids_d, mask_d = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
base = xl(ids_d, attention_mask=mask_d).last_hidden_state.clone()
nomask = xl(ids_d).last_hidden_state.clone()
eng.set_option("ws_poison", 1)
shown = 0
for it in range(150):
    out = xl(ids_d, attention_mask=mask_d).last_hidden_state
    nan_rows = torch.isnan(out).any(-1)
    d = (out - base).abs().amax(-1)
    bad = (d > 0) | nan_rows
    if bad.any() and shown < 6:
        shown += 1
        seqs = sorted(set(int(b) for b in torch.nonzero(bad)[:, 0].cpu()))
        msg = []
        for b in seqs[:6]:
            rows = torch.nonzero(bad[b])[:, 0].cpu().tolist()
            eq_nomask = bool(torch.equal(out[b], nomask[b]))
            dn = float((out[b] - nomask[b]).abs().max())
            msg.append(f"seq {b} len {int(lens[b])} rows {rows[0]}..{rows[-1]} ({len(rows)}) nan_rows {int(nan_rows[b].sum())} max|d| {float(d[b][~nan_rows[b]].max()) if (~nan_rows[b]).any() else -1:.3g} equals_nomask {eq_nomask} (max diff to nomask {dn:.3g})")
        print(f"it {it}: {len(seqs)} sequences differ:", " | ".join(msg))
print("done")
