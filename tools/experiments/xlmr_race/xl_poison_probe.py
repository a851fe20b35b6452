"""Debug probe: run-to-run differences of XLM-RoBERTa outputs (lanes, folding, poisoned workspace).  Runs on the library as committed; the
variants that need debug_scaffolding.patch are probe2-4.
python tools/xl_poison_probe.py lanes fold [calibrate B L]"""
import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from jegal_amd import synth
from jegal_amd._lib import Engine
from jegal_amd.xlmr import XLMRoberta
lanes, fold = int(sys.argv[1]), int(sys.argv[2])
cal = int(sys.argv[3]) if len(sys.argv) > 3 else 0
B = int(sys.argv[4]) if len(sys.argv) > 4 else 24
L = int(sys.argv[5]) if len(sys.argv) > 5 else 40
ids, mask = synth.xlmr_inputs(3, B, L)
eng = Engine(0)
eng.set_option("xlmr_lanes", lanes)
eng.set_option("xlmr_fold", fold)
for kv in os.environ.get("OPTS", "").split(","):
    if kv:
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
xl = XLMRoberta(engine=eng).load_state_dict(synth.xlmr_state_dict(layers=2))
ids_d, mask_d = torch.from_numpy(ids).cuda(), torch.from_numpy(mask).cuda()
if cal:
    xl.calibrate(ids_d, mask_d)
base = xl(ids_d, attention_mask=mask_d).last_hidden_state.clone()
n = 0
seqs = set()
eng.set_option("ws_poison", 1)
for it in range(150):
    eng.set_option("gemm_tile", it & 3)
    out = xl(ids_d, attention_mask=mask_d).last_hidden_state
    d = (out - base).abs().amax(-1).cpu().numpy()
    bad = np.argwhere(d > 0)
    if len(bad):
        n += 1
        seqs |= set(int(b) for b, _ in bad)
print(f"opts {os.environ.get('OPTS')} lanes {lanes} fold {fold} calibrated {cal} B {B} L {L}: runs with differences: {n} of 150; sequences {sorted(seqs)}")
